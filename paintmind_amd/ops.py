"""Operator-level host wrappers: torch tensors in, torch tensors out, compute in libpaintmind_hip.so.

PyTorch is plumbing here (device memory from its caching allocator, the current HIP stream).  Every
function asserts its operands live on a ROCm device; nothing falls back to ATen.
"""
import ctypes as C
import math

import torch

from . import _lib
from ._lib import BF16, F32, PART_K, PART_Q, PART_V, check

_TORCH2PM = {torch.float32: F32, torch.bfloat16: BF16}


def pm_dtype(dt):
    try:
        return _TORCH2PM[dt]
    except KeyError:
        raise TypeError(f"paintmind_amd computes in float32 or bfloat16, not {dt}") from None


def _dev(*tensors):
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise _lib.PmhipError(
                "paintmind_amd runs only on a ROCm device (got a CPU tensor); there is no CPU fallback")
        if not t.is_contiguous():
            raise ValueError("paintmind_amd ops need contiguous tensors")
        dev = t.device if dev is None else dev
        if t.device != dev:
            raise ValueError("operands live on different devices")
    return dev


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def stream_ptr(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def round_up(v, m):
    return (v + m - 1) // m * m


def swiglu_hidden(hidden_features):
    """hidden width rule of SwiGLUFFNFused (reference modules/mlp.py:53)."""
    return (int(hidden_features * 2 / 3) + 7) // 8 * 8


# ------------------------------------------------------------------------------------------------
def gemm(a, w, bias=None, residual=None, res_rows=0, out_dtype=None):
    """a[M,K] @ w[N,K]^T (+bias) (+residual[m % res_rows])."""
    dev = _dev(a, w, bias, residual)
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    out_dtype = out_dtype or a.dtype
    out = torch.empty(M, N, device=dev, dtype=out_dtype)
    ldr = residual.shape[-1] if residual is not None else 0
    rr = res_rows or (residual.shape[0] if residual is not None else 0)
    with torch.cuda.device(dev):
        check(lib.pmhip_gemm(pm_dtype(a.dtype), _p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(residual), ldr,
                             rr, _p(out), N, pm_dtype(out_dtype), M, N, K, stream_ptr(dev)), "pmhip_gemm")
    return out


def gemm_swiglu(a, w12p, b12p):
    dev = _dev(a, w12p, b12p)
    lib = _lib.load()
    M, K = a.shape
    Hp = w12p.shape[0] // 2
    out = torch.empty(M, Hp, device=dev, dtype=a.dtype)
    with torch.cuda.device(dev):
        check(lib.pmhip_gemm_swiglu(pm_dtype(a.dtype), _p(a), a.stride(0), _p(w12p), _p(b12p), _p(out), Hp, M, Hp, K,
                                    stream_ptr(dev)), "pmhip_gemm_swiglu")
    return out


def gemm_heads(a, w, heads, tokens, kinds, q_scale=1.0, dim_head=64):
    """Head-split projection; returns one tensor per part (Q [B,H,t,dh], K [B,H,tp,dh], V^T [B,H,dh,tp]).
    dim_head 64 runs the fused epilogue; other values the plain GEMM + split path (pmhip_gemm_heads_dh)."""
    dev = _dev(a, w)
    lib = _lib.load()
    M, K = a.shape
    B = M // tokens
    tp = round_up(tokens, 64)
    dh = int(dim_head)
    outs = []
    for kind in kinds:
        if kind == PART_Q:
            outs.append(torch.empty(B, heads, tokens, dh, device=dev, dtype=a.dtype))
        elif kind == PART_K:
            outs.append(torch.zeros(B, heads, tp, dh, device=dev, dtype=a.dtype))
        else:
            outs.append(torch.zeros(B, heads, dh, tp, device=dev, dtype=a.dtype))
    kinds_c = (C.c_int * len(kinds))(*kinds)
    outs_c = (C.c_void_p * len(kinds))(*[o.data_ptr() for o in outs])
    with torch.cuda.device(dev):
        if dh == 64:
            check(lib.pmhip_gemm_heads(pm_dtype(a.dtype), _p(a), a.stride(0), _p(w), w.stride(0), M, K, heads, tokens, tp,
                                       len(kinds), kinds_c, outs_c, float(q_scale), stream_ptr(dev)), "pmhip_gemm_heads")
        else:
            scratch = torch.empty(M, len(kinds) * heads * dh, device=dev, dtype=torch.float32)
            check(lib.pmhip_gemm_heads_dh(pm_dtype(a.dtype), _p(a), a.stride(0), _p(w), w.stride(0), M, K, heads, dh, tokens, tp,
                                          len(kinds), kinds_c, outs_c, float(q_scale), _p(scratch), stream_ptr(dev)),
                  "pmhip_gemm_heads_dh")
    return outs


# ---- bf16 hi/lo residual stream + LayerNorm folded into the consuming GEMM (include/pmhip.h, pmhip_lnfold) ------------
def split_hilo(x):
    """x f32 [M,D] -> (hi, lo) bf16 planes with x = hi + lo"""
    dev = _dev(x)
    lib = _lib.load()
    M, D = x.shape
    hi = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    lo = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    with torch.cuda.device(dev):
        check(lib.pmhip_split_hilo(_p(x), _p(hi), _p(lo), M, D, stream_ptr(dev)), "pmhip_split_hilo")
    return hi, lo


def join_hilo(hi, lo):
    dev = _dev(hi, lo)
    lib = _lib.load()
    M, D = hi.shape
    out = torch.empty(M, D, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        check(lib.pmhip_join_hilo(_p(hi), _p(lo), _p(out), M, D, stream_ptr(dev)), "pmhip_join_hilo")
    return out


def gemm_hilo(a, w, res_hi, res_lo, bias=None, res_rows=0, stats=False):
    """(hi, lo) = split(a @ w^T + bias + (res_hi + res_lo)[m % res_rows]); bf16 operands.
    stats=True: also the per-row partial statistics [M, N/64, 2] of the new hi plane (for ln_coef_parts)."""
    dev = _dev(a, w, res_hi, res_lo)
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    hi = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    lo = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    with torch.cuda.device(dev):
        if stats:
            parts = torch.full((M, N // 64, 2), float("nan"), device=dev, dtype=torch.float32)
            check(lib.pmhip_gemm_hilo_stats(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(res_hi), _p(res_lo), res_hi.stride(0),
                                            int(res_rows), _p(hi), _p(lo), N, M, N, K, _p(parts), stream_ptr(dev)), "pmhip_gemm_hilo_stats")
            return hi, lo, parts
        check(lib.pmhip_gemm_hilo(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(res_hi), _p(res_lo), res_hi.stride(0),
                                  int(res_rows), _p(hi), _p(lo), N, M, N, K, stream_ptr(dev)), "pmhip_gemm_hilo")
    return hi, lo


def gemm_hilo_center(a, w, res_hi, res_lo, bias=None, center_coef=None, shift=None, shift_mode=0, stats=False, center_extra=0.0):
    """gemm_hilo storing the pair of x - c, c = the row mean of the previous hi plane taken from `center_coef` ([M,2], ln_coef /
    ln_coef_parts); `shift` [M] fp32 is updated IN PLACE (shift_mode 1: <- 0, 2: += c).  -> (hi, lo[, parts])"""
    dev = _dev(a, w, res_hi, res_lo, center_coef, shift)
    lib = _lib.load()
    M, K = a.shape
    N = w.shape[0]
    hi = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    lo = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    parts = torch.full((M, N // 64, 2), float("nan"), device=dev, dtype=torch.float32) if stats else None
    with torch.cuda.device(dev):
        check(lib.pmhip_gemm_hilo_center(_p(a), a.stride(0), _p(w), w.stride(0), _p(bias), _p(res_hi), _p(res_lo), res_hi.stride(0), 0,
                                         _p(hi), _p(lo), N, M, N, K, _p(parts), _p(center_coef), float(center_extra), _p(shift), int(shift_mode),
                                         stream_ptr(dev)), "pmhip_gemm_hilo_center")
    return (hi, lo, parts) if stats else (hi, lo)


def unshift_hilo(hi, lo, shift):
    """in place: (hi, lo) <- split(hi + lo + shift[row])"""
    dev = _dev(hi, lo, shift)
    M, D = hi.shape
    with torch.cuda.device(dev):
        check(_lib.load().pmhip_unshift_hilo(_p(hi), _p(lo), _p(shift), M, D, stream_ptr(dev)), "pmhip_unshift_hilo")
    return hi, lo


def ln_coef_parts(parts, eps=1e-5):
    """per-row (rstd, -rstd * mean) from gemm_hilo(..., stats=True)'s partial statistics -> f32 [M,2]"""
    dev = _dev(parts)
    lib = _lib.load()
    M, nparts, _ = parts.shape
    coef = torch.empty(M, 2, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        check(lib.pmhip_ln_coef_parts(_p(parts), nparts, float(eps), _p(coef), M, stream_ptr(dev)), "pmhip_ln_coef_parts")
    return coef


def layernorm_hilo(hi, lo, gamma, beta, eps=1e-5, out_dtype=torch.bfloat16):
    dev = _dev(hi, lo, gamma, beta)
    lib = _lib.load()
    M, D = hi.shape
    out = torch.empty(M, D, device=dev, dtype=out_dtype)
    with torch.cuda.device(dev):
        check(lib.pmhip_layernorm_hilo(_p(hi), _p(lo), _p(gamma), _p(beta), float(eps), _p(out), pm_dtype(out_dtype), M, D,
                                       stream_ptr(dev)), "pmhip_layernorm_hilo")
    return out


def layernorm_to_hilo(x, gamma, beta, eps=1e-5):
    dev = _dev(x, gamma, beta)
    lib = _lib.load()
    M, D = x.shape
    hi = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    lo = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    with torch.cuda.device(dev):
        check(lib.pmhip_layernorm_to_hilo(_p(x), _p(gamma), _p(beta), float(eps), _p(hi), _p(lo), M, D, stream_ptr(dev)),
              "pmhip_layernorm_to_hilo")
    return hi, lo


def ln_coef(hi, eps=1e-5):
    """per-row (rstd, -rstd * mean) of the hi plane -> f32 [M,2]"""
    dev = _dev(hi)
    lib = _lib.load()
    M, D = hi.shape
    coef = torch.empty(M, 2, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        check(lib.pmhip_ln_coef(_p(hi), float(eps), _p(coef), M, D, stream_ptr(dev)), "pmhip_ln_coef")
    return coef


def _lnfold(coef, c, d, parts=None, eps=1e-5):
    """coef [M,2] is an input, or -- with `parts` [M, K/64, 2] (the producer's partial statistics) -- an OUTPUT the call fills"""
    ln = _lib.LnFold()
    ln.coef, ln.c, ln.d = coef.data_ptr(), c.data_ptr(), d.data_ptr()
    if parts is not None:
        if parts.dtype != torch.float32 or not parts.is_contiguous() or parts.shape[0] != coef.shape[0] or parts.shape[-1] != 2:
            raise ValueError("fold parts must be a contiguous fp32 [M, K/64, 2] tensor")
        ln.parts, ln.nparts, ln.eps = parts.data_ptr(), parts.shape[1], float(eps)
    ln._keep = (coef, c, d, parts)
    return ln


def lnfold_supported(kind, M, N, K):
    return bool(_lib.load().pmhip_lnfold_supported(BF16, kind, M, N, K))


def gemm_ln(xb, wg, coef, c, d, bias=None, out_dtype=None, parts=None):
    dev = _dev(xb, wg, coef, c, d)
    lib = _lib.load()
    M, K = xb.shape
    N = wg.shape[0]
    out_dtype = out_dtype or xb.dtype
    out = torch.empty(M, N, device=dev, dtype=out_dtype)
    ln = _lnfold(coef, c, d, parts)
    with torch.cuda.device(dev):
        check(lib.pmhip_gemm_ln(pm_dtype(xb.dtype), _p(xb), xb.stride(0), _p(wg), wg.stride(0), _p(bias), _p(out), N,
                                pm_dtype(out_dtype), M, N, K, C.byref(ln), stream_ptr(dev)), "pmhip_gemm_ln")
    return out


def gemm_softmax_stats(x, w, bias=None, fold=None):
    """logits f32 [M,N] = x @ w.T + bias and the softmax statistics of their 64-column blocks, f32 [M, N/64, 2] = (max, sum of exp)
    (pmhip_gemm_softmax_stats; transformer.py:91).  fold = (coef, c, d[, parts]): x is the hi plane and w the gamma-scaled weights
    of a folded LayerNorm (gemm_ln)."""
    dev = _dev(x, w, bias)
    lib = _lib.load()
    M, K = x.shape
    N = w.shape[0]
    if N % 64:
        raise ValueError("gemm_softmax_stats: N must be a multiple of 64")
    out = torch.empty(M, N, device=dev, dtype=torch.float32)
    stats = torch.empty(M, N // 64, 2, device=dev, dtype=torch.float32)
    ln = _lnfold(*fold) if fold is not None else None
    with torch.cuda.device(dev):
        check(lib.pmhip_gemm_softmax_stats(pm_dtype(x.dtype), _p(x), x.stride(0), _p(w), w.stride(0), _p(bias), _p(out), N, M, N, K,
                                           C.byref(ln) if ln is not None else None, _p(stats), stream_ptr(dev)),
              "pmhip_gemm_softmax_stats")
    return out, stats


def gemm_swiglu_ln(xb, w12pg, b12p, coef, c, d, parts=None):
    dev = _dev(xb, w12pg, b12p)
    lib = _lib.load()
    M, K = xb.shape
    Hp = w12pg.shape[0] // 2
    out = torch.empty(M, Hp, device=dev, dtype=xb.dtype)
    ln = _lnfold(coef, c, d, parts)
    with torch.cuda.device(dev):
        check(lib.pmhip_gemm_swiglu_ln(pm_dtype(xb.dtype), _p(xb), xb.stride(0), _p(w12pg), _p(b12p), _p(out), Hp, M, Hp, K,
                                       C.byref(ln), stream_ptr(dev)), "pmhip_gemm_swiglu_ln")
    return out


def gemm_heads_ln(xb, wg, heads, tokens, kinds, q_scale, coef, c, d, parts=None):
    dev = _dev(xb, wg)
    lib = _lib.load()
    M, K = xb.shape
    B = M // tokens
    tp = round_up(tokens, 64)
    outs = []
    for kind in kinds:
        shape = (B, heads, tokens, 64) if kind == PART_Q else ((B, heads, tp, 64) if kind == PART_K else (B, heads, 64, tp))
        outs.append(torch.empty(shape, device=dev, dtype=xb.dtype))
    kinds_c = (C.c_int * len(kinds))(*kinds)
    outs_c = (C.c_void_p * len(kinds))(*[o.data_ptr() for o in outs])
    ln = _lnfold(coef, c, d, parts)
    with torch.cuda.device(dev):
        check(lib.pmhip_gemm_heads_ln(pm_dtype(xb.dtype), _p(xb), xb.stride(0), _p(wg), wg.stride(0), M, K, heads, tokens, tp,
                                      len(kinds), kinds_c, outs_c, float(q_scale), C.byref(ln), stream_ptr(dev)),
              "pmhip_gemm_heads_ln")
    return outs


def attention_fallbacks(reset=False, device=None):
    """Workgroups of the bf16 attention kernel that had to be run again through its exact path since the last reset
    (a probability of the fixed-reference fast path left the f32 range; include/pmhip.h)."""
    lib = _lib.load()
    n = C.c_ulonglong(0)
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        check(lib.pmhip_attention_fallbacks(C.byref(n), int(bool(reset))), "pmhip_attention_fallbacks")
    return int(n.value)


def attention(q, k, vt, n_kv, use_exp2=False):
    """q [B,H,Nq,dh], k [B,H,Nkp,dh], vt [B,H,dh,Nkp] -> [B*Nq, H*dh]  (dh 64: tuned MFMA kernel; else pmhip_attention_dh)."""
    dev = _dev(q, k, vt)
    lib = _lib.load()
    B, H, Nq, dh = q.shape
    nkp = k.shape[2]
    out = torch.empty(B * Nq, H * dh, device=dev, dtype=q.dtype)
    with torch.cuda.device(dev):
        if dh == 64:
            check(lib.pmhip_attention(pm_dtype(q.dtype), _p(q), _p(k), _p(vt), _p(out), H * 64, B, H, Nq, n_kv, nkp,
                                      int(use_exp2), stream_ptr(dev)), "pmhip_attention")
        else:
            check(lib.pmhip_attention_dh(pm_dtype(q.dtype), _p(q), _p(k), _p(vt), _p(out), H * dh, B, H, dh, Nq, n_kv, nkp,
                                         int(use_exp2), stream_ptr(dev)), "pmhip_attention_dh")
    return out


def layernorm(x, gamma, beta, eps=1e-5, out_dtype=torch.float32):
    dev = _dev(x, gamma, beta)
    lib = _lib.load()
    M, D = x.shape
    out = torch.empty(M, D, device=dev, dtype=out_dtype)
    with torch.cuda.device(dev):
        check(lib.pmhip_layernorm(_p(x), _p(gamma), _p(beta), float(eps), _p(out), pm_dtype(out_dtype), M, D,
                                  stream_ptr(dev)), "pmhip_layernorm")
    return out


def patchify(img, patch, out_dtype=torch.float32):
    dev = _dev(img)
    lib = _lib.load()
    B, Cc, H, W = img.shape
    out = torch.empty(B * (H // patch) * (W // patch), Cc * patch * patch, device=dev, dtype=out_dtype)
    with torch.cuda.device(dev):
        check(lib.pmhip_patchify(_p(img), _p(out), pm_dtype(out_dtype), B, Cc, H, W, patch, stream_ptr(dev)),
              "pmhip_patchify")
    return out


def unpatchify_clamp(y, B, channels, size, patch, lo=-1.0, hi=1.0):
    dev = _dev(y)
    lib = _lib.load()
    img = torch.empty(B, channels, size, size, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        check(lib.pmhip_unpatchify_clamp(_p(y), _p(img), B, channels, size, size, patch, float(lo), float(hi),
                                         stream_ptr(dev)), "pmhip_unpatchify_clamp")
    return img


def convert_pad(x, kpad, out_dtype):
    dev = _dev(x)
    lib = _lib.load()
    M, K = x.shape
    out = torch.empty(M, kpad, device=dev, dtype=out_dtype)
    with torch.cuda.device(dev):
        check(lib.pmhip_convert_pad(_p(x), K, _p(out), pm_dtype(out_dtype), kpad, M, stream_ptr(dev)),
              "pmhip_convert_pad")
    return out


def add_rows(x, table):
    dev = _dev(x, table)
    lib = _lib.load()
    M, D = x.shape
    out = torch.empty_like(x)
    with torch.cuda.device(dev):
        check(lib.pmhip_add_rows(_p(x), _p(table), table.shape[0], _p(out), M, D, stream_ptr(dev)), "pmhip_add_rows")
    return out


def guidance_combine(cond, uncond, scale, out=None, with_stats=False):
    """uncond + scale * (cond - uncond), fp32, same shape; `out` may be one of the inputs.
    with_stats: returns (out, block_stats [..., V/64, 2]) -- the softmax statistics sample_rows(block_stats=) takes."""
    dev = _dev(cond, uncond)
    if cond.shape != uncond.shape or cond.dtype != torch.float32 or uncond.dtype != torch.float32:
        raise ValueError("guidance_combine needs two fp32 tensors of one shape")
    out = torch.empty(cond.shape, device=dev, dtype=torch.float32) if out is None else out
    if out.shape != cond.shape or out.dtype != torch.float32 or out.device != cond.device:
        raise ValueError("guidance_combine: `out` must be an fp32 tensor of the inputs' shape on their device")
    if not (cond.is_contiguous() and uncond.is_contiguous() and out.is_contiguous()):
        raise ValueError("guidance_combine needs contiguous tensors (the kernel walks them as flat arrays)")
    if cond.numel() % 4:
        raise ValueError("guidance_combine: the element count must be a multiple of 4")
    if with_stats:
        if cond.shape[-1] % 64:
            raise ValueError("guidance_combine(with_stats=True): rows must be a multiple of 64 long")
        stats = torch.empty(cond.shape[:-1] + (cond.shape[-1] // 64, 2), device=dev, dtype=torch.float32)
        with torch.cuda.device(dev):
            check(_lib.load().pmhip_guidance_combine_stats(_p(cond), _p(uncond), float(scale), _p(out), cond.numel(), _p(stats),
                                                           stream_ptr(dev)), "pmhip_guidance_combine_stats")
        return out, stats
    with torch.cuda.device(dev):
        check(_lib.load().pmhip_guidance_combine(_p(cond), _p(uncond), float(scale), _p(out), cond.numel(), stream_ptr(dev)),
              "pmhip_guidance_combine")
    return out


def embed_rows(table, ids, kpad, out_dtype):
    dev = _dev(table, ids)
    lib = _lib.load()
    V, E = table.shape
    M = ids.numel()
    out = torch.empty(M, kpad, device=dev, dtype=out_dtype)
    with torch.cuda.device(dev):
        check(lib.pmhip_embed_rows(_p(table), _p(ids), _p(out), pm_dtype(out_dtype), kpad, M, V, E, stream_ptr(dev)),
              "pmhip_embed_rows")
    return out


def vq_prepare(codebook):
    dev = _dev(codebook)
    lib = _lib.load()
    V, E = codebook.shape
    en = torch.empty_like(codebook)
    sq = torch.empty(V, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        check(lib.pmhip_vq_prepare(_p(codebook), _p(en), _p(sq), V, E, stream_ptr(dev)), "pmhip_vq_prepare")
    return en, sq


def vq_quantize(z, en, sq, beta=0.25):
    """z fp32 [M,E] -> (z_out [M,E], idx int64 [M], loss [1])."""
    dev = _dev(z, en, sq)
    lib = _lib.load()
    M, E = z.shape
    V = en.shape[0]
    z_out = torch.empty_like(z)
    idx = torch.empty(M, device=dev, dtype=torch.int64)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    scratch = torch.empty(lib.pmhip_vq_scratch_bytes(M, V), device=dev, dtype=torch.uint8)
    with torch.cuda.device(dev):
        check(lib.pmhip_vq_quantize(_p(z), _p(en), _p(sq), float(beta), _p(z_out), _p(idx), _p(loss), _p(scratch), M, V,
                                    E, stream_ptr(dev)), "pmhip_vq_quantize")
    return z_out, idx, loss


def sample_rows(logits, ids, mask_id, topk, temperature, noise=None, seed=0, step=0, row_base=0, block_stats=None):
    """logits fp32 [M,V], ids int64 [M] -> (pred [M], merged ids [M], score [M]).
    block_stats fp32 [M, V/64, 2] (gemm_softmax_stats / guidance_combine(with_stats=True)): the kernel reads them and the top-k
    blocks of a row instead of the row; same bits as without."""
    dev = _dev(logits, ids, noise)
    lib = _lib.load()
    M, V = logits.shape
    pred = torch.empty(M, device=dev, dtype=torch.int64)
    ids_out = torch.empty(M, device=dev, dtype=torch.int64)
    score = torch.empty(M, device=dev, dtype=torch.float32)
    if block_stats is not None:
        if (block_stats.dtype != torch.float32 or not block_stats.is_contiguous() or block_stats.device != logits.device
                or tuple(block_stats.shape) != (M, V // 64, 2) or V % 64):
            raise ValueError("sample_rows: block_stats must be a contiguous fp32 [M, V/64, 2] tensor on the logits' device")
        with torch.cuda.device(dev):
            check(lib.pmhip_sample_rows_stats(_p(logits), logits.stride(0), _p(block_stats), _p(ids), int(mask_id), int(topk),
                                              float(temperature), _p(noise), int(seed), int(step), int(row_base), _p(pred),
                                              _p(ids_out), _p(score), M, V, stream_ptr(dev)), "pmhip_sample_rows_stats")
        return pred, ids_out, score
    with torch.cuda.device(dev):
        check(lib.pmhip_sample_rows(_p(logits), logits.stride(0), _p(ids), int(mask_id), int(topk), float(temperature),
                                    _p(noise), int(seed), int(step), int(row_base), _p(pred), _p(ids_out), _p(score),
                                    M, V, stream_ptr(dev)), "pmhip_sample_rows")
    return pred, ids_out, score


def remask(ids, scores, num_mask, mask_id):
    """in place on ids int64 [B,N]."""
    dev = _dev(ids, scores)
    lib = _lib.load()
    B, N = ids.shape
    with torch.cuda.device(dev):
        check(lib.pmhip_remask(_p(ids), _p(scores), int(num_mask), int(mask_id), B, N, stream_ptr(dev)), "pmhip_remask")
    return ids


def random_mask(z, noise, mask_token, len_keep):
    """z fp32 [B,N,E], noise fp32 [B,N], mask_token fp32 [E] -> (x [B,N,E], mask [B,N] with 1 = masked)."""
    dev = _dev(z, noise, mask_token)
    lib = _lib.load()
    B, N, E = z.shape
    x = torch.empty_like(z)
    mask = torch.empty(B, N, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        check(lib.pmhip_random_mask(_p(z), _p(noise), _p(mask_token), int(len_keep), _p(x), _p(mask), B, N, E,
                                    stream_ptr(dev)), "pmhip_random_mask")
    return x, mask


def masked_ce(logits, labels, mask, label_smoothing=0.1):
    """logits fp32 [M,V], labels int64 [M], mask fp32 [M] -> (loss [1], row_loss [M])."""
    dev = _dev(logits, labels, mask)
    lib = _lib.load()
    M, V = logits.shape
    row_loss = torch.empty(M, device=dev, dtype=torch.float32)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    with torch.cuda.device(dev):
        check(lib.pmhip_masked_ce(_p(logits), logits.stride(0), _p(labels), _p(mask), float(label_smoothing), _p(row_loss),
                                  _p(loss), M, V, stream_ptr(dev)), "pmhip_masked_ce")
    return loss, row_loss


# ------------------------------------------------------------------------------------------------
def timing_enable(on=True):
    check(_lib.load().pmhip_timing_enable(int(on)))


def timing_reset():
    check(_lib.load().pmhip_timing_reset())


def timing_get(family):
    n = C.c_int(0)
    ms = C.c_double(0.0)
    check(_lib.load().pmhip_timing_get(family.encode(), C.byref(n), C.byref(ms)), "pmhip_timing_get")
    return n.value, ms.value


def device_info(device=0):
    cu = C.c_int(0)
    lds = C.c_int(0)
    arch = C.create_string_buffer(64)
    check(_lib.load().pmhip_device_info(int(device), C.byref(cu), C.byref(lds), arch, 64), "pmhip_device_info")
    return {"cu_count": cu.value, "lds_bytes": lds.value, "arch": arch.value.decode()}


LOG2E = math.log2(math.e)
__all__ = [n for n in dir() if not n.startswith("_")]
