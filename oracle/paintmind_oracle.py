"""CPU ORACLE -- test infrastructure, NOT product code.

A plain numpy (float32) restatement of the reference's generation path, used only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker / baseline.  Nothing under
paintmind_amd/ imports this file.

Pinning: the reference has no tests or golden vectors of its own, so this oracle
is pinned against outputs of the reference itself, captured in the development container by
tests/golden/make_goldens.py (which imports /root/reference) and committed under tests/golden/.
tests/test_oracle_vs_golden.py checks every function below against those vectors.

Parameters are addressed by the reference's state_dict key names; `p` is a dict
name -> float32 ndarray.  Citations are reference file:line.
"""
import math

import numpy as np

F = np.float32


def _f(a):
    return np.ascontiguousarray(a, dtype=F)


# ------------------------------------------------------------------------------------------------
# primitives
# ------------------------------------------------------------------------------------------------
def linear(x, w, b=None):
    """torch.nn.Linear: x @ w^T + b."""
    y = x @ w.T
    return y + b if b is not None else y


def layernorm(x, g, b, eps=1e-5):
    """torch.nn.LayerNorm (biased variance, eps inside the sqrt)."""
    mu = x.mean(-1, keepdims=True, dtype=F)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True, dtype=F)
    return xc / np.sqrt(var + F(eps)) * g + b


def softmax(x):
    e = np.exp(x - x.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True, dtype=F)


def silu(x):
    return x / (F(1) + np.exp(-x))


def swiglu_hidden(h):
    """modules/mlp.py:53"""
    return (int(h * 2 / 3) + 7) // 8 * 8


def attention(x, context, p, pre, heads):
    """CrossAttention.forward, modules/attention.py:43-59. x [B,N,D], context [B,L,Dc] or None."""
    B, N, _ = x.shape
    c = x if context is None else context                      # :47
    q = linear(x, p[pre + "to_q.weight"])                      # :46
    k = linear(c, p[pre + "to_k.weight"])                      # :48
    v = linear(c, p[pre + "to_v.weight"])                      # :49
    dh = q.shape[-1] // heads
    split = lambda t: t.reshape(B, t.shape[1], heads, dh).transpose(0, 2, 1, 3)   # 'b n (h d) -> (b h) n d' :51
    q, k, v = split(q), split(k), split(v)
    q = q * F(dh ** -0.5)                                      # :52
    sim = softmax(q @ k.transpose(0, 1, 3, 2))                 # :54-55
    out = (sim @ v).transpose(0, 2, 1, 3).reshape(B, N, heads * dh)   # :57-58
    return linear(out, p[pre + "to_out.0.weight"], p[pre + "to_out.0.bias"])   # :59


def swiglu_ffn(x, p, pre):
    """SwiGLUFFN.forward, modules/mlp.py:27-31."""
    x12 = linear(x, p[pre + "w12.weight"], p[pre + "w12.bias"])
    h = x12.shape[-1] // 2
    hidden = silu(x12[..., :h]) * x12[..., h:]
    return linear(hidden, p[pre + "w3.weight"], p[pre + "w3.bias"])


def layer_stage1(x, p, pre, heads):
    """stage1/layers.py:54-58"""
    x = attention(layernorm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"]), None, p, pre + "attn1.", heads) + x
    return swiglu_ffn(layernorm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"]), p, pre + "ffnet.") + x


def layer_stage2(x, context, p, pre, heads):
    """stage2/transformer.py:44-49"""
    x = attention(layernorm(x, p[pre + "norm1.weight"], p[pre + "norm1.bias"]), None, p, pre + "attn1.", heads) + x
    x = attention(layernorm(x, p[pre + "norm2.weight"], p[pre + "norm2.bias"]), context, p, pre + "attn2.", heads) + x
    return swiglu_ffn(layernorm(x, p[pre + "norm3.weight"], p[pre + "norm3.bias"]), p, pre + "ffnet.") + x


# ------------------------------------------------------------------------------------------------
# stage 1
# ------------------------------------------------------------------------------------------------
def patchify(img, patch):
    """rows = patches in (h w) order, columns in (c, kh, kw) order = flattened Conv2d weight
    (stage1/layers.py:81-84: Conv2d k=s=patch, then 'b c h w -> b (h w) c')."""
    B, C, H, W = img.shape
    g = H // patch
    x = img.reshape(B, C, g, patch, W // patch, patch).transpose(0, 2, 4, 1, 3, 5)
    return x.reshape(B, g * (W // patch), C * patch * patch)


def unpatchify(y, channels, size, patch):
    """'b (h w) (p1 p2 c) -> b c (h p1) (w p2)', stage1/layers.py:150 (c is the fastest axis)."""
    B = y.shape[0]
    g = size // patch
    x = y.reshape(B, g, g, patch, patch, channels).transpose(0, 5, 1, 3, 2, 4)
    return x.reshape(B, channels, size, size)


def encoder(img, p, cfg, pre="encoder.", collect=None):
    """Encoder.forward, stage1/layers.py:106-112"""
    e = cfg["enc"]
    w = p[pre + "to_patch_embedding.0.weight"].reshape(e["dim"], -1)
    x = patchify(_f(img), e["patch_size"]) @ w.T                                  # :107
    x = x + p[pre + "position_embedding"]                                         # :108
    x = layernorm(x, p[pre + "norm_pre.weight"], p[pre + "norm_pre.bias"])        # :109
    for i in range(e["depth"]):                                                   # :110
        x = layer_stage1(x, p, f"{pre}transformer.layers.{i}.", e["num_head"])
        if collect is not None:
            collect.append(x)
    return x


def decoder(x, p, cfg, pre="decoder."):
    """Decoder.forward, stage1/layers.py:145-152 (un-clamped)"""
    d = cfg["dec"]
    x = x + p[pre + "position_embedding"]                                         # :146
    for i in range(d["depth"]):                                                   # :147
        x = layer_stage1(x, p, f"{pre}transformer.layers.{i}.", d["num_head"])
    x = layernorm(x, p[pre + "norm.weight"], p[pre + "norm.bias"])                # :148
    x = linear(x, p[pre + "proj.weight"], p[pre + "proj.bias"])                   # :149
    return unpatchify(x, d["out_channels"], d["image_size"], d["patch_size"])     # :150


def l2norm(t):
    """F.normalize(t, p=2, dim=-1), stage1/quantize.py:5-6 (eps 1e-12)"""
    n = np.sqrt((t * t).sum(-1, keepdims=True, dtype=F))
    return t / np.maximum(n, F(1e-12))


def vq_distances(zn_flat, en):
    """stage1/quantize.py:24-26: (sum z^2 + sum e^2) - 2 z.e^T"""
    return ((zn_flat * zn_flat).sum(1, keepdims=True, dtype=F) + (en * en).sum(1, dtype=F)) - F(2) * (zn_flat @ en.T)


def vq_forward(z, codebook, beta=0.25):
    """VectorQuantizer.forward, stage1/quantize.py:18-38 -> (z + (z_q - z), loss, indices)"""
    zn = l2norm(_f(z))                                                            # :19
    flat = zn.reshape(-1, zn.shape[-1])
    en = l2norm(codebook)                                                         # :21
    d = vq_distances(flat, en)
    idx = d.argmin(1).reshape(zn.shape[:-1])                                      # :28 first minimum
    zq = l2norm(codebook[idx])                                                    # :29-30
    diff2 = ((zq - zn) ** 2).mean(dtype=F)
    loss = F(beta) * diff2 + diff2                                                # :33
    return zn + (zq - zn), loss, idx.astype(np.int64)                             # :36


def vq_decode_indices(idx, codebook):
    """VectorQuantizer.decode_from_indice, stage1/quantize.py:40-44"""
    return l2norm(codebook[idx])


def vqgan_encode(img, p, cfg, return_pre=False):
    """VQModel.encode, stage1/vqmodel.py:21-25"""
    x = encoder(img, p, cfg)
    ze = linear(x, p["prev_quant.weight"], p["prev_quant.bias"])
    out = vq_forward(ze, p["quantize.embedding.weight"], cfg["beta"])
    return out + (ze,) if return_pre else out


def vqgan_decode(z, p, cfg):
    """VQModel.decode, stage1/vqmodel.py:27-30"""
    x = linear(_f(z), p["post_quant.weight"], p["post_quant.bias"])
    return np.clip(decoder(x, p, cfg), -1.0, 1.0).astype(F)


def vqgan_decode_indices(idx, p, cfg):
    """VQModel.decode_from_indice, stage1/vqmodel.py:38-41"""
    return vqgan_decode(vq_decode_indices(idx, p["quantize.embedding.weight"]), p, cfg)


# ------------------------------------------------------------------------------------------------
# stage 2
# ------------------------------------------------------------------------------------------------
def cond_transformer(tokens, context, p, s2cfg, pre="transformer."):
    """CondTransformer.forward, stage2/transformer.py:80-93"""
    x = linear(_f(tokens), p[pre + "token_proj.weight"], p[pre + "token_proj.bias"])     # :81
    x = x + p[pre + "position_embedding"]                                                # :82
    if context is not None and (pre + "context_proj.weight") in p:                       # :84-85
        context = linear(_f(context), p[pre + "context_proj.weight"])
    for i in range(s2cfg["depth"]):                                                      # :87-88
        x = layer_stage2(x, context, p, f"{pre}layers.layer{i}.", s2cfg["num_head"])
    x = layernorm(x, p[pre + "norm.weight"], p[pre + "norm.bias"])                       # :90
    return linear(x, p[pre + "to_logits.weight"], p[pre + "to_logits.bias"])             # :91


def ids2tokens(ids, p, vq_pre="vqgan."):
    """Pipeline.ids2tokens, generate.py:148-157: RAW codebook rows + mask token"""
    table = np.concatenate([p[vq_pre + "quantize.embedding.weight"], p["mask_token"]], 0)
    return table[ids]


def mask_schedule(ratio):
    """generate.py:25-26 (float64)"""
    return np.cos(math.pi / 2. * ratio)


def num_token_masked(mask_ratio, num_tokens):
    """generate.py:175"""
    return max(int(mask_ratio * num_tokens), 1)


def gumbel_from_uniform(u):
    """generate.py:29-30,40-42: -log(-log(u)) with both logs clamped at 1e-20"""
    inner = -np.log(np.maximum(u, F(1e-20)))
    return -np.log(np.maximum(inner, F(1e-20)))


def order_desc_then_index(values):
    """The tie rule of this build: larger value first, equal values by smaller index (stable)."""
    return np.argsort(-values.astype(np.float64), axis=-1, kind="stable")


def sample_rows(logits, ids, mask_id, topk, temperature, noise):
    """generate.py:163-173 on flattened rows.  logits [M,V] f32, ids [M], noise [M,V] uniform(0,1).
    -> pred [M], merged ids [M], score [M]"""
    logits = _f(logits)
    M, V = logits.shape
    order = order_desc_then_index(logits)[:, :topk]                       # top_k, :33-37
    filt = np.full_like(logits, -np.inf)
    np.put_along_axis(filt, order, np.take_along_axis(logits, order, 1), 1)
    pert = filt / F(max(temperature, 1e-10)) + gumbel_from_uniform(_f(noise))    # :45-46
    pred = pert.argmax(1)                                                 # first maximum
    is_mask = ids == mask_id                                              # :166
    merged = np.where(is_mask, pred, ids)                                 # :168
    probs = softmax(logits)                                               # :170 (UNfiltered)
    score = F(1) - np.take_along_axis(probs, pred[:, None], 1)[:, 0]      # :171
    score = np.where(is_mask, score, F(-1e5)).astype(F)                   # :173
    return pred.astype(np.int64), merged.astype(np.int64), score


def remask(ids, scores, num_mask, mask_id):
    """generate.py:177-179: ids[b, topk(scores[b], num_mask)] = mask_id (ties: smaller index first)"""
    ids = ids.copy()
    sel = order_desc_then_index(scores)[:, :num_mask]
    np.put_along_axis(ids, sel, mask_id, 1)
    return ids


def remask_threshold_sets(scores, num_mask):
    """For tie-tolerant comparison with torch.topk (implementation-defined tie order): per image the
    set that MUST be selected (score strictly above the num_mask-th largest) and the set that MAY be."""
    srt = -np.sort(-scores, axis=-1)
    thr = srt[:, num_mask - 1:num_mask]
    return scores > thr, scores >= thr


def sample_step(ids, mask_ratio, context, topk, temperature, noise, p, cfg, s2cfg, decode=True, guidance_scale=None):
    """Pipeline.sample, generate.py:159-181 -> (ids', img, aux).
    guidance_scale (not in the reference's sampling; the reference only TRAINS the unconditional branch, utils/trainer.py:379,
    387-388): logits = uncond + scale * (cond - uncond) with uncond = the same forward with context=None, one fused multiply-add
    per element in float32 -- intended behaviour, unpinned by the reference."""
    B, N = ids.shape
    mask_id = cfg["n_embed"]
    tok = ids2tokens(ids, p)
    logits = cond_transformer(tok, context, p, s2cfg)                                      # :161-162
    if guidance_scale is not None:
        uncond = cond_transformer(tok, None, p, s2cfg)
        diff = (logits - uncond).astype(np.float32)
        logits = (np.float64(np.float32(guidance_scale)) * diff.astype(np.float64) + uncond.astype(np.float64)).astype(np.float32)
    V = logits.shape[-1]
    pred, merged, score = sample_rows(logits.reshape(B * N, V), ids.reshape(-1), mask_id, topk, temperature,
                                      noise.reshape(B * N, V))
    vq_p = {k[len("vqgan."):]: v for k, v in p.items() if k.startswith("vqgan.")}
    img = vqgan_decode_indices(pred.reshape(B, N), vq_p, cfg) if decode else None          # :165 (pred at ALL positions)
    nm = num_token_masked(mask_ratio, N)                                                   # :175
    new_ids = remask(merged.reshape(B, N), score.reshape(B, N), nm, mask_id)               # :177-179
    aux = {"logits": logits, "pred": pred.reshape(B, N), "merged": merged.reshape(B, N), "score": score.reshape(B, N),
           "num_mask": nm}
    return new_ids, img, aux


def region_keep_mask(coord, image_size, patch_size, keep_inside):
    """generate.py:204-208 (inpaint: ones with the rectangle zeroed) / :223-227 (outpaint: zeros with the rectangle
    set).  coord = (x, y, h, w) in pixels, floor-divided by the patch size; note the reference slices rows with
    y:y+h and columns with x:x+w.  -> bool [1, grid*grid], True = token is kept."""
    s = patch_size
    x, y, h, w = coord[0] // s, coord[1] // s, coord[2] // s, coord[3] // s
    g = image_size // s
    keep = np.zeros((g, g), dtype=bool) if keep_inside else np.ones((g, g), dtype=bool)
    keep[y:y + h, x:x + w] = keep_inside
    return keep.reshape(1, -1)


def region_loop(img, coord, context, timesteps, topk, temperature, p, cfg, s2cfg, keep_inside, noises=None, ids=None):
    """Pipeline.inpaint (keep_inside=False, generate.py:200-217) / Pipeline.outpaint (True, :219-236):
    encode -> ids kept outside / inside the rectangle, mask id elsewhere -> `timesteps` sample steps -> last image.

    The reference computes `ids * mask + mask_token_id * (1 - mask)` in float (:210,229), which nn.Embedding then
    rejects (tests/golden/api.json: inpaint_runs == false); this is that expression with the evident integer result.
    `noises`: one uniform(0,1) array [B,N,V] per step (irrelevant for topk=1); `ids`: start from given token ids
    instead of encoding (used to compare the composition on identical tokens).  -> (img, ids, aux of the last step)"""
    vq_p = {k[len("vqgan."):]: v for k, v in p.items() if k.startswith("vqgan.")}
    if ids is None:
        _, _, ids = vqgan_encode(img, vq_p, cfg)                                            # to_latent, :125-131
    e = cfg["enc"]
    keep = region_keep_mask(coord, e["image_size"], e["patch_size"], keep_inside)
    ids = np.where(keep, ids, cfg["n_embed"]).astype(np.int64)
    out, aux = None, None
    for step in range(timesteps):
        r = mask_schedule((step + 1) / timesteps)
        cur_temp = temperature * (1 - step / timesteps)
        noise = noises[step] if noises is not None else np.full(ids.shape + (cfg["n_embed"],), 0.5, dtype=F)
        ids, out, aux = sample_step(ids, r, context, topk, cur_temp, noise, p, cfg, s2cfg)
    return out, ids, aux


# ------------------------------------------------------------------------------------------------
# masked-token objective, forward only (generate.py:78-146)
# ------------------------------------------------------------------------------------------------
def random_masking(x, mask_token, mask_ratio, noise):
    """generate.py:78-110 given the uniforms of :89.  -> (x', mask) with mask 1 = replaced by mask_token"""
    B, L, D = x.shape
    len_keep = L - max(int(L * mask_ratio), 1)                                              # :86-87
    order = np.argsort(noise, axis=1, kind="stable")                                        # :92 (ascending)
    rank = np.argsort(order, axis=1, kind="stable")                                         # :93
    mask = (rank >= len_keep).astype(F)                                                     # :103-107
    out = np.where(mask[..., None] > 0, _f(mask_token).reshape(1, 1, D), _f(x))             # :95-101
    return out.astype(F), mask


def masked_ce(logits, labels, masks, label_smoothing=0.1):
    """generate.py:112-125: F.cross_entropy(label_smoothing, reduction='none') * mask, / mask.sum()"""
    x = logits.reshape(-1, logits.shape[-1]).astype(np.float64)
    y = labels.reshape(-1)
    m = masks.reshape(-1).astype(np.float64)
    mx = x.max(1, keepdims=True)
    logp = x - mx - np.log(np.exp(x - mx).sum(1, keepdims=True))
    nll = -logp[np.arange(len(y)), y]
    smooth = -logp.mean(1)
    row = (1 - label_smoothing) * nll + label_smoothing * smooth
    return F((row * m).sum() / m.sum()), (row * m).astype(F)


def pipeline_forward(img, context, mask_ratio, noise, p, cfg, s2cfg):
    """Pipeline.forward, generate.py:136-146 -> (loss, aux)"""
    vq_p = {k[len("vqgan."):]: v for k, v in p.items() if k.startswith("vqgan.")}
    z, _, idx = vqgan_encode(img, vq_p, cfg)
    x, mask = random_masking(z, p["mask_token"], mask_ratio, noise)
    logits = cond_transformer(x, context, p, s2cfg)
    loss, row = masked_ce(logits, idx, mask)
    return loss, {"x": x, "mask": mask, "logits": logits, "row_loss": row, "ids": idx}


# ------------------------------------------------------------------------------------------------
# Philox4x32-10, the counter-based generator of the perf-mode sampler (paintmind_amd/csrc/sample.hip)
# ------------------------------------------------------------------------------------------------
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c = [np.asarray(v, dtype=np.uint64) & 0xFFFFFFFF for v in (c0, c1, c2, c3)]
    k0 = np.uint64(k0) & np.uint64(0xFFFFFFFF)
    k1 = np.uint64(k1) & np.uint64(0xFFFFFFFF)
    m0, m1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = m0 * c[0]
        p1 = m1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c = [(hi1 ^ c[1] ^ k0) & mask, lo1, (hi0 ^ c[3] ^ k1) & mask, lo0]
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return c


def philox_uniform(seed, step, global_rows, cols):
    """u in [0,1) for (row, col) pairs exactly as sample.hip draws them (24-bit mantissa)."""
    rows = np.asarray(global_rows, dtype=np.uint64)
    x = philox4x32_10(rows & np.uint64(0xFFFFFFFF), rows >> np.uint64(32), np.asarray(cols, dtype=np.uint64),
                      np.uint64(step), np.uint64(seed) & np.uint64(0xFFFFFFFF), np.uint64(seed) >> np.uint64(32))[0]
    return ((x >> np.uint64(8)).astype(np.float64) * (1.0 / 16777216.0)).astype(F)
