"""Seeded randomised differential tests: HIP ops through the C ABI against the CPU oracle on shapes nobody picked by
hand (ragged M / N, odd tile counts, every epilogue, ragged attention, arbitrary class counts for the samplers).
Bit-exact for index outputs, the op-level tolerances of test_gpu_ops.py for floating point."""
import numpy as np
import pytest
import torch

from gpu_common import bf16_round, dev, n, rel_err, t
from oracle import paintmind_oracle as O
from paintmind_amd import ops, packing

pytestmark = pytest.mark.gpu


def _rng(seed):
    return np.random.default_rng(1000 + seed)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_gemm(seed):
    r = _rng(seed)
    dtype = torch.bfloat16 if seed % 2 else torch.float32
    M = int(r.choice([1, 7, 64, 129, 255, 256, 300, 1000, 2048, 4096 + 17, 8192]))
    N = int(r.integers(1, 130)) * 8
    K = int(r.integers(1, 17)) * 64
    if seed >= 12:                                      # big, tile-aligned: the 256x256 / two-workgroup kernels
        M, N, K = 256 * int(r.integers(40, 130)), 128 * int(r.integers(2, 9)), 64 * int(r.integers(1, 25))
        dtype = torch.bfloat16
    a = r.standard_normal((M, K)).astype(np.float32)
    w = (r.standard_normal((N, K)) * K ** -0.5).astype(np.float32)
    if dtype == torch.bfloat16:
        a, w = bf16_round(a), bf16_round(w)
    bias = r.standard_normal(N).astype(np.float32) if r.random() < 0.7 else None
    use_res = r.random() < 0.6
    res_rows = int(r.choice([1, 16, M])) if use_res else None
    res = r.standard_normal((res_rows, N)).astype(np.float32) if use_res else None
    out_dtype = torch.float32 if (use_res or r.random() < 0.5) else dtype
    ref = a.astype(np.float64) @ w.astype(np.float64).T
    if bias is not None:
        ref += bias
    if use_res:
        ref += res[np.arange(M) % res_rows]
    out = n(ops.gemm(t(a, dtype), t(w, dtype), bias=None if bias is None else t(bias), residual=None if res is None else t(res),
                     res_rows=res_rows or 0, out_dtype=out_dtype))
    tol = 3e-5 if out_dtype == torch.float32 else 1e-2
    assert out.shape == (M, N) and rel_err(out, ref) < tol, (M, N, K, dtype, out_dtype, rel_err(out, ref))


@pytest.mark.parametrize("seed", range(8))
def test_fuzz_swiglu_and_heads(seed):
    r = _rng(100 + seed)
    dtype = torch.bfloat16 if seed % 2 else torch.float32
    D = int(r.choice([64, 128, 512, 768]))
    H = int(r.integers(3, 200)) * 8
    M = int(r.choice([5, 256, 777, 4096, 16384])) if seed < 6 else 256 * int(r.integers(48, 100))
    lin = torch.nn.Linear(D, 2 * H)
    x = r.standard_normal((M, D)).astype(np.float32)
    if dtype == torch.bfloat16:
        x = bf16_round(x)
        lin.weight.data = torch.from_numpy(bf16_round(lin.weight.detach().numpy()))
    w, b = lin.weight.detach().numpy().astype(np.float64), lin.bias.detach().numpy().astype(np.float64)
    x12 = x.astype(np.float64) @ w.T + b
    ref = x12[:, :H] / (1 + np.exp(-x12[:, :H])) * x12[:, H:]
    w12p, b12p, hp = packing.pack_w12(lin.to(dev()), dtype)
    out = n(ops.gemm_swiglu(t(x, dtype), w12p, b12p))
    assert out.shape == (M, hp) and np.all(out[:, H:] == 0)
    assert rel_err(out[:, :H], ref) < (3e-5 if dtype == torch.float32 else 2e-2)
    # head-split projection of the same rows, viewed as B images of `tokens` tokens
    tokens = int(r.choice([tk for tk in (1, 5, 16, 64, 77, 256, 1024) if M % tk == 0]))
    B, heads = M // tokens, int(r.integers(1, 5))
    wqkv = (r.standard_normal((3 * heads * 64, D)) * D ** -0.5).astype(np.float32)
    if dtype == torch.bfloat16:
        wqkv = bf16_round(wqkv)
    q, k, vt = ops.gemm_heads(t(x, dtype), t(wqkv, dtype), heads, tokens, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.37)
    full = (x.astype(np.float64) @ wqkv.astype(np.float64).T).reshape(B, tokens, 3, heads, 64)
    tol = 3e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(n(q), full[:, :, 0].transpose(0, 2, 1, 3) * 0.37) < tol
    assert rel_err(n(k)[:, :, :tokens], full[:, :, 1].transpose(0, 2, 1, 3)) < tol
    assert rel_err(n(vt)[:, :, :, :tokens], full[:, :, 2].transpose(0, 2, 3, 1)) < tol


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_attention(seed):
    r = _rng(200 + seed)
    fast = bool(seed % 2)
    dtype = torch.bfloat16 if fast else torch.float32
    B, H = int(r.integers(1, 4)), int(r.integers(1, 5))
    Nq, Nkv = int(r.integers(1, 700)), int(r.integers(1, 700))
    nkp = -(-Nkv // 64) * 64
    q = r.standard_normal((B, H, Nq, 64)).astype(np.float32) * float(r.choice([0.3, 1.0, 4.0]))
    k = r.standard_normal((B, H, Nkv, 64)).astype(np.float32)
    v = r.standard_normal((B, H, Nkv, 64)).astype(np.float32)
    scale = 0.125
    if fast:
        k, v = bf16_round(k), bf16_round(v)
        qs = bf16_round(q * (scale * ops.LOG2E))
        s = (qs.astype(np.float64) @ k.astype(np.float64).transpose(0, 1, 3, 2)) * np.log(2.0)
    else:
        qs = q * np.float32(scale)
        s = qs.astype(np.float64) @ k.astype(np.float64).transpose(0, 1, 3, 2)
    s -= s.max(-1, keepdims=True)
    pr = np.exp(s)
    pr /= pr.sum(-1, keepdims=True)
    ref = (pr @ v.astype(np.float64)).transpose(0, 2, 1, 3).reshape(B * Nq, H * 64)
    kp = np.full((B, H, nkp, 64), np.nan, np.float32)             # padding must never be read into the result
    kp[:, :, :Nkv] = k
    vtp = np.full((B, H, 64, nkp), np.nan, np.float32)
    vtp[:, :, :, :Nkv] = v.transpose(0, 1, 3, 2)
    out = n(ops.attention(t(qs, dtype), t(kp, dtype), t(vtp, dtype), Nkv, use_exp2=fast))
    assert np.isfinite(out).all() and rel_err(out, ref) < (5e-5 if not fast else 4e-2), (B, H, Nq, Nkv, rel_err(out, ref))


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_sampling_tail(seed):
    r = _rng(300 + seed)
    V = int(r.integers(2, 2200)) * 4
    M = int(r.integers(1, 300))
    topk = int(r.integers(1, min(64, V) + 1))
    temp = float(r.choice([0.0, 0.3, 1.0, 2.5]))
    logits = (r.standard_normal((M, V)) * float(r.choice([0.5, 3.0, 20.0]))).astype(np.float32)
    if seed % 3 == 0:                                              # ties: quantised logits
        logits = np.round(logits * 2) / 2
    ids = r.integers(0, V + 1, M).astype(np.int64)                 # V = mask id
    noise = r.random((M, V)).astype(np.float32)
    pred, merged, score = ops.sample_rows(t(logits), t(ids), V, topk, temp, noise=t(noise))
    pr, mr, sr = O.sample_rows(logits, ids, V, topk, temp, noise)
    assert np.array_equal(n(pred), pr) and np.array_equal(n(merged), mr)
    assert np.allclose(n(score), sr, rtol=1e-4, atol=1e-6)
    B, N = int(r.integers(1, 6)), int(r.integers(2, 1500))
    scores = r.random((B, N)).astype(np.float32)
    scores[r.random((B, N)) < 0.3] = -1e5
    if seed % 2:
        scores = np.round(scores, 1)                               # ties in the confidence ranking
    ids2 = r.integers(0, 50, (B, N)).astype(np.int64)
    m = int(r.integers(1, N + 1))
    assert np.array_equal(n(ops.remask(t(ids2.copy()), t(scores), m, 999)), O.remask(ids2, scores, m, 999))


@pytest.mark.parametrize("seed", range(6))
def test_fuzz_rows(seed):
    r = _rng(400 + seed)
    M, D = int(r.integers(1, 3000)), int(r.integers(1, 400)) * 4
    x = (r.standard_normal((M, D)) * 3 + r.standard_normal((M, 1))).astype(np.float32)
    g, b = r.standard_normal(D).astype(np.float32), r.standard_normal(D).astype(np.float32)
    assert rel_err(n(ops.layernorm(t(x), t(g), t(b), 1e-5, torch.float32)), O.layernorm(x, g, b)) < 2e-5
    assert rel_err(n(ops.layernorm(t(x), t(g), t(b), 1e-5, torch.bfloat16)), O.layernorm(x, g, b)) < 1e-2
    Bn, N, E = int(r.integers(1, 5)), int(r.integers(1, 1200)), int(r.integers(1, 17)) * 4
    z = r.standard_normal((Bn, N, E)).astype(np.float32)
    noise = r.random((Bn, N)).astype(np.float32)
    if seed % 2:
        noise = np.round(noise, 2)                                 # ties: stable order decides
    tok = r.standard_normal(E).astype(np.float32)
    ratio = float(r.choice([0.0, 0.15, 0.5, 0.9, 1.0]))
    xm, mask = ops.random_mask(t(z), t(noise), t(tok), N - max(int(N * ratio), 1))
    xo, mo = O.random_masking(z, tok, ratio, noise)
    assert np.array_equal(n(mask), mo) and np.array_equal(n(xm), xo)
    V = int(r.integers(2, 2100)) * 4
    Mr = int(r.integers(1, 500))
    logits = (r.standard_normal((Mr, V)) * 3).astype(np.float32)
    labels = r.integers(0, V, Mr)
    msk = (r.random(Mr) < 0.6).astype(np.float32)
    msk[0] = 1
    loss, rows = ops.masked_ce(t(logits), t(labels), t(msk), 0.1)
    lo, ro = O.masked_ce(logits, labels, msk, 0.1)
    assert abs(float(loss) - float(lo)) < 1e-4 and np.abs(n(rows) - ro).max() < 1e-4


@pytest.mark.parametrize("seed", range(10))
def test_fuzz_gemm_hilo_center(seed):
    """pmhip_gemm_hilo_center at random shapes over every kernel route: the centred pair + the accumulated shift reproduce the
    float64 x_new; the shift equals the previous plane's row mean + center_extra; a second producer chained on the first keeps
    the identity (the shift accumulates)."""
    r = _rng(400 + seed)
    bf = torch.bfloat16
    if seed < 4:
        M, N, K = int(r.choice([1, 100, 300, 777, 2048])), 64 * int(r.integers(1, 9)), 64 * int(r.integers(1, 9))
    else:
        M, N, K = 256 * int(r.integers(24, 100)), int(r.choice([256, 512, 768, 1024])), 64 * int(r.integers(1, 25))
    a = bf16_round(r.standard_normal((M, K)).astype(np.float32) * 0.7)
    w = bf16_round((r.standard_normal((N, K)) * K ** -0.5).astype(np.float32))
    b0 = r.standard_normal(N).astype(np.float32)
    off = float(r.choice([0.0, 3.0, 40.0]))
    res = (r.standard_normal((M, N)) * 1.2 + off + r.standard_normal((M, 1))).astype(np.float32)
    rh, rl = ops.split_hilo(t(res))
    x = n(ops.join_hilo(rh, rl)).astype(np.float64)
    shift = torch.zeros(M, device=dev())
    extra = float(np.float32(b0.mean()))
    tot_shift = np.zeros(M)
    hi, lo = rh, rl
    for rep in range(2):
        coef = ops.ln_coef(hi)
        cen = -n(coef)[:, 1].astype(np.float64) / n(coef)[:, 0].astype(np.float64) + extra
        x = a.astype(np.float64) @ w.astype(np.float64).T + b0 + x
        hi, lo = ops.gemm_hilo_center(t(a, bf), t(w, bf), hi, lo, bias=t(b0), center_coef=coef, shift=shift, shift_mode=2, center_extra=extra)
        tot_shift += cen
        sh = n(shift).astype(np.float64)
        assert np.abs(sh - tot_shift).max() < 1e-4 * max(1.0, np.abs(tot_shift).max()), (rep, M, N, K)
        got = n(ops.join_hilo(hi, lo)).astype(np.float64) + sh[:, None]
        scale = max(1.0, np.abs(x - sh[:, None]).max())
        assert np.abs(got - x).max() < 4e-5 * scale * (rep + 1), (rep, M, N, K, np.abs(got - x).max())
        x = got                                              # chain on what the planes really hold
