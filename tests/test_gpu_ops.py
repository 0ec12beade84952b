"""Operator-level parity on the MI355X: every C-ABI op against the CPU oracle on the same seeded inputs.
fp32 ops: tolerance 1e-5 relative (fp32 accumulation order differs); integer outputs bit-exact.
bf16 ops: inputs are rounded to bf16 on both sides, tolerance 2e-2 relative to the output scale."""
import numpy as np
import pytest
import torch

from gpu_common import bf16_round, dev, n, rel_err, t
from oracle import paintmind_oracle as O
from oracle import vq_ref
from paintmind_amd import _lib, ops, packing
from paintmind_amd._lib import PmhipError
from util import maxabs

pytestmark = pytest.mark.gpu
RNG = np.random.default_rng(1234)


def rnd(*shape, scale=1.0):
    return (RNG.standard_normal(shape) * scale).astype(np.float32)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 512, 512), (300, 192, 192), (1024, 32, 512), (77, 8192, 128), (2048, 2048, 1408)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gemm_bias_residual(M, N, K, dtype):
    a, w, b, r = rnd(M, K), rnd(N, K, scale=K ** -0.5), rnd(N), rnd(16, N)
    if dtype == torch.bfloat16:
        a, w = bf16_round(a), bf16_round(w)
    ref = a @ w.T + b + r[np.arange(M) % 16]
    out = ops.gemm(t(a, dtype), t(w, dtype), bias=t(b), residual=t(r), res_rows=16, out_dtype=torch.float32)
    assert rel_err(n(out), ref) < (2e-5 if dtype == torch.float32 else 2e-5), rel_err(n(out), ref)
    out2 = ops.gemm(t(a, dtype), t(w, dtype))          # no epilogue terms, output in the compute dtype
    tol = 2e-5 if dtype == torch.float32 else 1e-2
    assert rel_err(n(out2), a @ w.T) < tol


@pytest.mark.parametrize("M,N,K", [(8192, 2048, 512), (16384, 1024, 192), (12288, 4096, 64), (65536, 512, 1408)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_large_tile_kernel(M, N, K, out_dtype):
    """shapes that take the 256x256 phase-staggered kernel (M, N multiples of 256, >= 192 tiles); K tiles: 8, 3 (odd), 1, 22"""
    a, w, b = bf16_round(rnd(M, K)), bf16_round(rnd(N, K, scale=K ** -0.5)), rnd(N)
    out = n(ops.gemm(t(a, torch.bfloat16), t(w, torch.bfloat16), bias=t(b), out_dtype=out_dtype))
    ref = a @ w.T + b
    assert rel_err(out, ref) < (2e-5 if out_dtype == torch.float32 else 1e-2), rel_err(out, ref)
    # every row / column block is touched: compare block-wise maxima too
    blk = np.abs(out - ref).reshape(M // 256, 256, N // 256, 256).max(axis=(1, 3))
    assert blk.max() < (1e-3 if out_dtype == torch.float32 else 0.15 * np.abs(ref).max())


def test_gemm_large_tile_kernel_with_residual():
    """K >= 1024 residual GEMMs (the SwiGLU w3 shape) also take the 256x256 kernel; residual rows are tiled (res_rows)"""
    M, N, K = 65536, 512, 1408
    a, w, b = bf16_round(rnd(M, K)), bf16_round(rnd(N, K, scale=K ** -0.5)), rnd(N)
    r = rnd(M, N)
    out = n(ops.gemm(t(a, torch.bfloat16), t(w, torch.bfloat16), bias=t(b), residual=t(r), out_dtype=torch.float32))
    assert rel_err(out, a @ w.T + b + r) < 2e-5
    pos = rnd(1024, N)
    out = n(ops.gemm(t(a, torch.bfloat16), t(w, torch.bfloat16), bias=t(b), residual=t(pos), res_rows=1024, out_dtype=torch.float32))
    assert rel_err(out, a @ w.T + b + pos[np.arange(M) % 1024]) < 2e-5


@pytest.mark.parametrize("M,N,K", [(65536, 512, 512), (32768, 768, 768), (49152, 1024, 64)])
def test_gemm_two_workgroup_kernel_short_k_residual(M, N, K):
    """f32-out residual GEMMs with K < 1024 and >= 192 tiles of 256x128 (the attention out-projections at bench batch
    sizes) take gemm2b.hip; full residual and row-modulo residual (position embedding) against the oracle"""
    a, w, b = bf16_round(rnd(M, K)), bf16_round(rnd(N, K, scale=K ** -0.5)), rnd(N)
    r = rnd(M, N)
    out = n(ops.gemm(t(a, torch.bfloat16), t(w, torch.bfloat16), bias=t(b), residual=t(r), out_dtype=torch.float32))
    ref = a @ w.T + b
    assert rel_err(out, ref + r) < 2e-5
    assert np.abs(out - (ref + r)).reshape(M // 256, 256, N // 128, 128).max(axis=(1, 3)).max() < 1e-3
    pos = rnd(1024, N)
    out = n(ops.gemm(t(a, torch.bfloat16), t(w, torch.bfloat16), bias=t(b), residual=t(pos), res_rows=1024, out_dtype=torch.float32))
    assert rel_err(out, ref + pos[np.arange(M) % 1024]) < 2e-5
    out = n(ops.gemm(t(a, torch.bfloat16), t(w, torch.bfloat16), residual=t(r), out_dtype=torch.float32))      # no bias
    assert rel_err(out, a @ w.T + r) < 2e-5


def test_swiglu_and_heads_large_tile_kernel():
    M, D, H = 16384, 512, 1368
    lin = torch.nn.Linear(D, 2 * H)
    x = bf16_round(rnd(M, D))
    lin.weight.data = torch.from_numpy(bf16_round(lin.weight.detach().numpy()))
    w, b = lin.weight.detach().numpy(), lin.bias.detach().numpy()
    x12 = x @ w.T + b
    ref = O.silu(x12[:, :H]) * x12[:, H:]
    w12p, b12p, hp = packing.pack_w12(lin.to(dev()), torch.bfloat16)
    out = n(ops.gemm_swiglu(t(x, torch.bfloat16), w12p, b12p))
    assert rel_err(out[:, :H], ref) < 2e-2 and np.all(out[:, H:] == 0)
    # head-split projection, B=16 images of 1024 tokens, 8 heads
    B, heads, N = 16, 8, 1024
    wqkv = bf16_round(rnd(3 * heads * 64, D, scale=D ** -0.5))
    q, k, vt = ops.gemm_heads(t(x, torch.bfloat16), t(wqkv, torch.bfloat16), heads, N, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.125)
    full = (x @ wqkv.T).reshape(B, N, 3, heads, 64)
    assert rel_err(n(q), full[:, :, 0].transpose(0, 2, 1, 3) * 0.125) < 1e-2
    assert rel_err(n(k), full[:, :, 1].transpose(0, 2, 1, 3)) < 1e-2
    assert rel_err(n(vt), full[:, :, 2].transpose(0, 2, 3, 1)) < 1e-2


def test_hilo_row_operators():
    """x = hi + lo as two bf16 planes: the split is exact to 2^-17 |x|, joins back, and LayerNorm of the pair equals LayerNorm of x"""
    bf = torch.bfloat16
    for M, D in ((300, 512), (64, 768), (129, 1024), (7, 64), (5, 72)):
        x = rnd(M, D, scale=3.0) + 0.9
        hi, lo = ops.split_hilo(t(x))
        assert torch.equal(hi, t(x).to(bf))                                   # hi IS bf16(x): what the next GEMM multiplies
        back = n(ops.join_hilo(hi, lo))
        assert np.max(np.abs(back - x) / np.abs(x).max()) < 2.0 ** -16
        gamma, beta = 1 + 0.3 * rnd(D), 0.2 * rnd(D)
        x64 = back.astype(np.float64)
        y64 = (x64 - x64.mean(1, keepdims=True)) / np.sqrt(x64.var(1, keepdims=True) + 1e-5) * gamma + beta
        assert maxabs(n(ops.layernorm_hilo(hi, lo, t(gamma), t(beta), out_dtype=torch.float32)), y64) < 2e-5
        assert rel_err(n(ops.layernorm_hilo(hi, lo, t(gamma), t(beta), out_dtype=bf)), y64) < 1e-2
        yh, yl = ops.layernorm_to_hilo(t(x), t(gamma), t(beta))
        want = n(ops.layernorm(t(x), t(gamma), t(beta)))
        assert maxabs(n(ops.join_hilo(yh, yl)), want) < 1e-4
        assert float((yh.float() - t(want).to(bf).float()).abs().max()) <= 2.0 ** -7 * float(np.abs(want).max())   # one bf16 ulp
        # coefficients of the folded consumer: statistics of the hi plane
        h64 = n(hi.float()).astype(np.float64)
        rstd = 1.0 / np.sqrt(h64.var(1) + 1e-5)
        coef = n(ops.ln_coef(hi))
        assert rel_err(coef[:, 0], rstd) < 1e-5 and maxabs(coef[:, 1], -rstd * h64.mean(1)) < 1e-4


@pytest.mark.parametrize("M,N,K", [(2048, 512, 512), (8192, 768, 768), (4096, 512, 2048), (300, 128, 64), (33 * 1024, 1024, 1024), (1024, 192, 64)])
def test_gemm_hilo_row_statistics_and_coefficients(M, N, K):
    """pmhip_gemm_hilo_stats: the producer's epilogue also leaves, per row and 64-column part, (sum, sum of squares about the
    part's mean) of the NEW hi plane; pmhip_ln_coef_parts combines them into the folded LayerNorm's (rstd, -rstd * mean).  Same
    planes as the plain producer bit for bit; statistics against float64 of the hi plane; coefficients equal to the plane pass
    (pmhip_ln_coef) to a few ulp.  Shapes cover every GEMM route (256x128 two-workgroup, 256x256, 128x128) and ragged M."""
    bf = torch.bfloat16
    a, w, b0 = bf16_round(rnd(M, K, scale=0.7)), bf16_round(rnd(N, K, scale=K ** -0.5)), rnd(N)
    res = rnd(M, N) * 1.5 + 0.4
    rh, rl = ops.split_hilo(t(res))
    hi0, lo0 = ops.gemm_hilo(t(a, bf), t(w, bf), rh, rl, bias=t(b0))
    hi, lo, parts = ops.gemm_hilo(t(a, bf), t(w, bf), rh, rl, bias=t(b0), stats=True)
    assert torch.equal(hi, hi0) and torch.equal(lo, lo0)
    h64 = hi.double().cpu().numpy().reshape(M, N // 64, 64)
    p = parts.cpu().numpy()
    assert np.isfinite(p).all()
    assert np.abs(p[..., 0] - h64.sum(-1)).max() < 2e-4
    assert np.abs(p[..., 1] - ((h64 - h64.mean(-1, keepdims=True)) ** 2).sum(-1)).max() < 2e-3
    coef = n(ops.ln_coef_parts(parts))
    want = n(ops.ln_coef(hi))
    hfull = hi.double().cpu().numpy()
    rstd64 = 1.0 / np.sqrt(hfull.var(-1) + 1e-5)
    assert np.abs(coef[:, 0] / rstd64 - 1).max() < 2e-6 and np.abs(coef[:, 1] + rstd64 * hfull.mean(-1)).max() < 2e-6
    assert np.abs(coef / want - 1).max() < 4e-6, np.abs(coef / want - 1).max()
    again = ops.gemm_hilo(t(a, bf), t(w, bf), rh, rl, bias=t(b0), stats=True)[2]
    assert torch.equal(again, parts)                                  # deterministic (no atomics)


@pytest.mark.parametrize("M,N,K,rows", [(65536 // 8, 512, 512, 0),      # two-workgroup kernel (short-K residual GEMM)
                                         (8192, 512, 1408, 0),          # 256x256 kernel (long K)
                                         (2048, 512, 64, 1024),         # 128x128 kernel, position-embedding addend (row modulo)
                                         (200, 768, 192, 0)])           # ragged M
def test_gemm_hilo_residual_stream(M, N, K, rows):
    """(hi, lo) <- split(A W^T + bias + (hi + lo)): every kernel route, against float64; in place like the engine uses it"""
    bf = torch.bfloat16
    a, w, b0 = bf16_round(rnd(M, K)), bf16_round(rnd(N, K, scale=K ** -0.5)), rnd(N)
    R = rows or M
    res = rnd(R, N, scale=2.0) + 0.5
    rh, rl = ops.split_hilo(t(res))
    r64 = n(ops.join_hilo(rh, rl)).astype(np.float64)
    want = a.astype(np.float64) @ w.astype(np.float64).T + b0 + r64[np.arange(M) % R]
    hi, lo = ops.gemm_hilo(t(a, bf), t(w, bf), rh, rl, bias=t(b0), res_rows=rows)
    got = n(ops.join_hilo(hi, lo)).astype(np.float64)
    assert np.max(np.abs(got - want)) < 2e-5 * max(1.0, np.abs(want).max()), np.max(np.abs(got - want))
    assert bool((lo.float().abs() <= hi.float().abs() * 2.0 ** -8 + 1e-30).all())     # lo is the rounding remainder of hi
    # same value as the fp32-stream kernel up to the pair's 2^-17 resolution
    plain = n(ops.gemm(t(a, bf), t(w, bf), bias=t(b0), residual=ops.join_hilo(rh, rl), res_rows=rows, out_dtype=torch.float32))
    assert np.max(np.abs(got - plain)) < 2.0 ** -15 * np.abs(plain).max()


def test_layernorm_fold_consumers():
    """LayerNorm folded into the GEMM that consumes it (stage1/layers.py:54-58: every projection consumes LN(x)): the hi plane
    times gamma-scaled weights + the epilogue formula with pmhip_ln_coef's coefficients, against the float64
    LayerNorm -> Linear of the represented x, and against the unfolded bf16 kernels (plain / head split / SwiGLU)."""
    M, D, heads = 4096, 512, 8
    bf = torch.bfloat16
    x = rnd(M, D) + 0.7                                     # non-zero row means
    hi, lo = ops.split_hilo(t(x))
    coef = ops.ln_coef(hi)
    gamma, beta = 1 + 0.3 * rnd(D), 0.2 * rnd(D)
    x64 = n(ops.join_hilo(hi, lo)).astype(np.float64)
    y64 = (x64 - x64.mean(1, keepdims=True)) / np.sqrt(x64.var(1, keepdims=True) + 1e-5) * gamma + beta
    y_dev = ops.layernorm_hilo(hi, lo, t(gamma), t(beta), out_dtype=bf)
    # plain consumer (N = 1536 -> 96 tiles of 256x256), f32 result with bias
    w1, b1 = bf16_round(rnd(1536, D, scale=D ** -0.5)), rnd(1536)
    wg, c, d = packing.ln_fold(t(w1), t(gamma), t(beta))
    got = n(ops.gemm_ln(hi, wg, coef, c, d, bias=t(b1), out_dtype=torch.float32))
    want = y64 @ w1.astype(np.float64).T + b1
    unfused = n(ops.gemm(y_dev, t(w1, bf), bias=t(b1), out_dtype=torch.float32))
    assert rel_err(got, want) < 2e-2 and rel_err(unfused, want) < 2e-2, (rel_err(got, want), rel_err(unfused, want))
    # served at ANY tile count (folded or not may not depend on the batch size): 256 rows alone give the same bits
    assert ops.lnfold_supported(0, M, 1536, D) and ops.lnfold_supported(0, 256, 1536, D)
    assert not ops.lnfold_supported(0, 200, 1536, D) and not ops.lnfold_supported(0, M, 192, D)
    small = ops.gemm_ln(hi[:256].contiguous(), wg, coef[:256].contiguous(), c, d, bias=t(b1), out_dtype=torch.float32)
    assert np.array_equal(n(small), got[:256])
    # head split (q | k | v)
    q, k, vt = ops.gemm_heads_ln(hi, wg, heads, 1024, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.125, coef, c, d)
    q2, k2, vt2 = ops.gemm_heads(y_dev, t(w1, bf), heads, 1024, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.125)
    proj = (y64 @ w1.astype(np.float64).T).reshape(4, 1024, 3, heads, 64)
    assert rel_err(n(q), proj[:, :, 0].transpose(0, 2, 1, 3) * 0.125) < 2e-2
    assert rel_err(n(k), proj[:, :, 1].transpose(0, 2, 1, 3)) < 2e-2 and rel_err(n(vt), proj[:, :, 2].transpose(0, 2, 3, 1)) < 2e-2
    assert rel_err(n(q), n(q2)) < 2e-2 and rel_err(n(vt), n(vt2)) < 2e-2
    # SwiGLU (hidden 1368 -> 1408 padded, packed rows)
    lin = torch.nn.Linear(D, 2 * 1368)
    w12p32, b12p, hp = packing.pack_w12(lin.to(dev()), torch.float32)
    wg, c, d = packing.ln_fold(w12p32, t(gamma), t(beta))
    hid = n(ops.gemm_swiglu_ln(hi, wg, b12p, coef, c, d))
    hid2 = n(ops.gemm_swiglu(y_dev, w12p32.to(bf), b12p))
    w12, b12 = n(lin.weight).astype(np.float64), n(lin.bias).astype(np.float64)
    x12 = y64 @ w12.T + b12
    ref = O.silu(x12[:, :1368].astype(np.float32)).astype(np.float64) * x12[:, 1368:]
    assert rel_err(hid[:, :1368], ref) < 3e-2 and rel_err(hid2[:, :1368], ref) < 3e-2 and np.all(hid[:, 1368:] == 0)


def test_folded_gemm_of_a_small_batch_is_bit_identical_to_the_large_batch_kernel():
    """Round 5: a folded GEMM whose 256x256 tiling would be at most 128 workgroups (B = 1: 24 for q|k|v, 44 for SwiGLU, 128 for
    the logits) runs on the 128x128 kernel with the same fold in its epilogue.  Same MFMA chain per element, same epilogue
    arithmetic: one image alone (small kernel) must equal the same image inside a batch of eight (256x256 kernel) bit for
    bit, for the three epilogues."""
    B, T, D, heads = 8, 1024, 512, 8
    M = B * T
    x = rnd(M, D) + 0.4
    hi, lo = ops.split_hilo(t(x))
    coef = ops.ln_coef(hi)
    gamma, beta = 1 + 0.3 * rnd(D), 0.2 * rnd(D)
    hi1, coef1 = hi[:T].contiguous(), coef[:T].contiguous()
    # plain consumer, f32 out with bias (the logits shape: N = 8192 -> 8 * 4 * 32 = 1024 tiles vs 128)
    w1, b1 = bf16_round(rnd(8192, D, scale=D ** -0.5)), rnd(8192)
    wg, c, d = packing.ln_fold(t(w1), t(gamma), t(beta))
    big = ops.gemm_ln(hi, wg, coef, c, d, bias=t(b1), out_dtype=torch.float32)
    one = ops.gemm_ln(hi1, wg, coef1, c, d, bias=t(b1), out_dtype=torch.float32)
    assert torch.equal(one, big[:T])
    # head split q | k | v (192 tiles vs 24)
    w2 = bf16_round(rnd(1536, D, scale=D ** -0.5))
    wg, c, d = packing.ln_fold(t(w2), t(gamma), t(beta))
    kinds = [ops.PART_Q, ops.PART_K, ops.PART_V]
    qb, kb, vb = ops.gemm_heads_ln(hi, wg, heads, T, kinds, 0.125 * ops.LOG2E, coef, c, d)
    q1, k1, v1 = ops.gemm_heads_ln(hi1, wg, heads, T, kinds, 0.125 * ops.LOG2E, coef1, c, d)
    assert torch.equal(q1[0], qb[0]) and torch.equal(k1[0], kb[0]) and torch.equal(v1[0], vb[0])
    # SwiGLU (352 tiles vs 44)
    lin = torch.nn.Linear(D, 2 * 1368)
    w12p32, b12p, hp = packing.pack_w12(lin.to(dev()), torch.float32)
    wg, c, d = packing.ln_fold(w12p32, t(gamma), t(beta))
    assert torch.equal(ops.gemm_swiglu_ln(hi1, wg, b12p, coef1, c, d), ops.gemm_swiglu_ln(hi, wg, b12p, coef, c, d)[:T])
    # coefficients from the producer's partial statistics (pmhip_lnfold::parts): the small launch computes them in its own prologue
    # and writes them out, the large one has pmhip_ln_coef_parts launched in front of it -- same coefficients, same GEMM bits
    a, w3, b3 = bf16_round(rnd(M, 256, scale=0.7)), bf16_round(rnd(D, 256, scale=1 / 16)), rnd(D)
    nh, nl, parts = ops.gemm_hilo(t(a, torch.bfloat16), t(w3, torch.bfloat16), hi, lo, bias=t(b3), stats=True)
    want = ops.ln_coef_parts(parts)
    nan = lambda rows: torch.full((rows, 2), float("nan"), device=dev())
    cb, c1 = nan(M), nan(T)
    big = ops.gemm_swiglu_ln(nh, wg, b12p, cb, c, d, parts=parts)
    one = ops.gemm_swiglu_ln(nh[:T].contiguous(), wg, b12p, c1, c, d, parts=parts[:T].contiguous())
    assert torch.equal(cb, want) and torch.equal(c1, want[:T]) and torch.equal(one, big[:T])
    assert torch.equal(big, ops.gemm_swiglu_ln(nh, wg, b12p, want, c, d))


@pytest.mark.parametrize("M,N,K", [(8192, 512, 512),       # two-workgroup kernel
                                   (8192, 512, 1408),      # 256x256 kernel
                                   (300, 128, 64)])        # 128x128 kernel, ragged M (unpipelined epilogue)
def test_gemm_hilo_center(M, N, K):
    """pmhip_gemm_hilo_center: the producer stores the pair of x - c, c = the row mean of the PREVIOUS hi plane (from the
    (rstd, -rstd * mean) pairs of pmhip_ln_coef), and accumulates c into `shift`.  hi + lo + shift reproduces the float64 result
    like the plain producer does; the new hi plane is centred (the rows carry a common offset of 30); its statistics describe
    the centred plane; pmhip_unshift_hilo folds the shift back in; shift_mode 1 opens a stream (shift <- 0)."""
    bf = torch.bfloat16
    a, w, b0 = bf16_round(rnd(M, K, scale=0.7)), bf16_round(rnd(N, K, scale=K ** -0.5)), rnd(N)
    res = rnd(M, N) * 1.5 + 30.0 + 3.0 * rnd(M, 1)
    rh, rl = ops.split_hilo(t(res))
    r64 = n(ops.join_hilo(rh, rl)).astype(np.float64)
    want = a.astype(np.float64) @ w.astype(np.float64).T + b0 + r64
    coef = ops.ln_coef(rh) if N % 8 == 0 else None
    cen = -n(coef)[:, 1].astype(np.float64) / n(coef)[:, 0].astype(np.float64)
    assert np.abs(cen - n(rh.float()).astype(np.float64).mean(1)).max() < 1e-3
    shift = torch.full((M,), 7.0, device=dev())
    stats = N % 64 == 0
    out = ops.gemm_hilo_center(t(a, bf), t(w, bf), rh, rl, bias=t(b0), center_coef=coef, shift=shift, shift_mode=2, stats=stats)
    hi, lo = out[0], out[1]
    sh = n(shift).astype(np.float64)
    assert np.abs(sh - 7.0 - cen).max() < 1e-4                                    # shift += c
    got = n(ops.join_hilo(hi, lo)).astype(np.float64) + sh[:, None] - 7.0
    assert np.max(np.abs(got - want)) < 3e-5 * max(1.0, np.abs(want - cen[:, None]).max()), np.max(np.abs(got - want))
    assert np.abs(n(hi.float()).mean(1)).max() < 2.0 and np.abs(want.mean(1)).min() > 15.0     # centred plane, offset rows
    assert bool((lo.float().abs() <= hi.float().abs() * 2.0 ** -8 + 1e-30).all())
    if stats:
        h64 = hi.double().cpu().numpy().reshape(M, N // 64, 64)
        p = out[2].cpu().numpy()
        assert np.abs(p[..., 0] - h64.sum(-1)).max() < 2e-4
        assert np.abs(p[..., 1] - ((h64 - h64.mean(-1, keepdims=True)) ** 2).sum(-1)).max() < 2e-3
    # back to the plain pair
    sh_t = shift - 7.0
    uh, ul = ops.unshift_hilo(hi.clone(), lo.clone(), sh_t)
    back = n(ops.join_hilo(uh, ul)).astype(np.float64)
    assert np.max(np.abs(back - want) / np.abs(want).max()) < 2.0 ** -15
    assert bool((ul.float().abs() <= uh.float().abs() * 2.0 ** -8 + 1e-30).all())  # a plain pair again: lo is the remainder of hi
    # no centring + shift_mode 1: the plain producer's planes bit for bit, shift zeroed
    hi0, lo0 = ops.gemm_hilo(t(a, bf), t(w, bf), rh, rl, bias=t(b0))
    hi1, lo1 = ops.gemm_hilo_center(t(a, bf), t(w, bf), rh, rl, bias=t(b0), shift=shift, shift_mode=1)
    assert torch.equal(hi0, hi1) and torch.equal(lo0, lo1) and float(shift.abs().max()) == 0.0
    again = ops.gemm_hilo_center(t(a, bf), t(w, bf), rh, rl, bias=t(b0), center_coef=coef)
    assert torch.equal(again[0], hi) and torch.equal(again[1], lo)               # deterministic; shift optional
    # center_extra (the mean of the bias: the part of the new row mean known in advance) is subtracted and accounted too
    shift.fill_(0.0)
    hx, lx = ops.gemm_hilo_center(t(a, bf), t(w, bf), rh, rl, bias=t(b0 + 20.0), center_coef=coef, shift=shift, shift_mode=2,
                                  center_extra=float(np.float32(b0.mean() + 20.0)))
    gotx = n(ops.join_hilo(hx, lx)).astype(np.float64) + n(shift).astype(np.float64)[:, None]
    assert np.max(np.abs(gotx - (want + 20.0))) < 3e-5 * max(1.0, np.abs(want - cen[:, None]).max())
    assert np.abs(n(hx.float()).mean(1) - n(hi.float()).mean(1) + b0.mean()).max() < 0.05    # the extra offset never reaches the plane


@pytest.mark.parametrize("offset,massive", [(0.0, 0), (2.0, 0), (2.0, 3), (50.0, 3)])
def test_layernorm_fold_accuracy_with_row_offsets_and_massive_channels(offset, massive):
    """The folded LayerNorm normalises bf16(x) (the hi plane): x is rounded BEFORE the mean is subtracted, so a row whose common
    offset is large against its spread loses 2^-9 |x| / std per element where the unfolded pmhip_layernorm_hilo (statistics and
    normalisation of hi + lo in fp32, one rounding of the result) loses 2^-9 |LN(x)|.  Measured here for rows with
    mean = offset x std and `massive` outlier channels of 100 x std -- the shape of trained transformers' activations; the seeded
    random-init weights of the parity suite have offset ~ 0.  Bars: the fold stays within 1.5x of the unfolded kernel's error up
    to offset 2; at offset 50 its error must stay inside the rounding model 2^-9 * offset (a regression detector: such
    checkpoints should run with PMHIP_LN_UNFOLD=1)."""
    M, D, N = 2048, 512, 1536
    bf = torch.bfloat16
    x = rnd(M, D) + offset * (1.0 + 0.2 * rnd(M, 1))
    if massive:
        x[:, :massive] += 100.0 * np.sign(rnd(1, massive))
    hi, lo = ops.split_hilo(t(x))
    gamma, beta = 1 + 0.3 * rnd(D), 0.2 * rnd(D)
    x64 = n(ops.join_hilo(hi, lo)).astype(np.float64)
    y64 = (x64 - x64.mean(1, keepdims=True)) / np.sqrt(x64.var(1, keepdims=True) + 1e-5) * gamma + beta
    w1 = bf16_round(rnd(N, D, scale=D ** -0.5))
    want = y64 @ w1.astype(np.float64).T
    wg, c, d = packing.ln_fold(t(w1), t(gamma), t(beta))
    got = n(ops.gemm_ln(hi, wg, ops.ln_coef(hi), c, d, out_dtype=torch.float32))
    unf = n(ops.gemm(ops.layernorm_hilo(hi, lo, t(gamma), t(beta), out_dtype=bf), t(w1, bf), out_dtype=torch.float32))
    scale = np.abs(want).std()
    e_fold, e_unf = np.abs(got - want).mean() / scale, np.abs(unf - want).mean() / scale
    print(f"offset {offset} x std, {massive} massive channels: mean |err| / std(out): folded {e_fold:.5f}, unfolded {e_unf:.5f}")
    if offset <= 2.0 and not massive:
        assert e_fold < 3.0 * e_unf + 1e-3, (e_fold, e_unf)
    # rounding model: per-element error of bf16(x) relative to the row's spread, averaged by the K = 512 contraction
    rel = 2.0 ** -9 * max(1.0, np.abs(x).mean() / x.std(1).mean())
    assert e_fold < 4.0 * rel + 4e-3, (e_fold, rel)
    assert e_unf < 6e-3, e_unf


def test_gemm_is_transpose_correct_on_asymmetric_data():
    """A = identity-like selector, W asymmetric: catches a swapped (m,n) in the MFMA output mapping."""
    M = N = K = 128
    a = np.eye(M, K, dtype=np.float32)
    w = (np.arange(N)[:, None] * 1000 + np.arange(K)[None, :]).astype(np.float32)
    out = n(ops.gemm(t(a), t(w)))
    assert np.array_equal(out, w.T)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("D,H", [(512, 1368), (64, 88), (768, 2048)])
def test_swiglu(dtype, D, H):
    M = 200
    lin = torch.nn.Linear(D, 2 * H)
    x = rnd(M, D)
    w, b = lin.weight.detach().numpy(), lin.bias.detach().numpy() + rnd(2 * H, scale=0.1)
    lin.bias.data = torch.from_numpy(b)
    if dtype == torch.bfloat16:
        x, w = bf16_round(x), bf16_round(w)
        lin.weight.data = torch.from_numpy(w)
    x12 = x @ w.T + b
    ref = O.silu(x12[:, :H]) * x12[:, H:]
    lin = lin.to(dev())
    w12p, b12p, hp = packing.pack_w12(lin, dtype)
    out = n(ops.gemm_swiglu(t(x, dtype), w12p, b12p))
    assert out.shape == (M, hp)
    assert np.all(out[:, H:] == 0)                       # padded hidden columns are exactly zero
    assert rel_err(out[:, :H], ref) < (2e-5 if dtype == torch.float32 else 2e-2)


def _attention_ref(q, k, v, heads, scale):
    B, Nq, inner = q.shape
    dh = inner // heads
    sp = lambda x: x.reshape(B, x.shape[1], heads, dh).transpose(0, 2, 1, 3)
    s = O.softmax((sp(q) * np.float32(scale)) @ sp(k).transpose(0, 1, 3, 2))
    return (s @ sp(v)).transpose(0, 2, 1, 3).reshape(B * Nq, inner)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,heads,Nq,Nkv", [(2, 2, 128, 128), (1, 8, 1024, 1024), (2, 3, 16, 16), (2, 4, 1024, 77), (1, 2, 200, 130)])
def test_heads_projection_and_attention(dtype, B, heads, Nq, Nkv):
    D = 128
    inner = heads * 64
    x, c = rnd(B * Nq, D), rnd(B * Nkv, D)
    wq, wk, wv = (rnd(inner, D, scale=D ** -0.5) for _ in range(3))
    if dtype == torch.bfloat16:
        x, c, wq, wk, wv = map(bf16_round, (x, c, wq, wk, wv))
    fast = dtype == torch.bfloat16
    scale = 0.125
    (q,) = ops.gemm_heads(t(x, dtype), t(wq, dtype), heads, Nq, [ops.PART_Q], scale * (ops.LOG2E if fast else 1.0))
    k, vt = ops.gemm_heads(t(c, dtype), t(np.concatenate([wk, wv]), dtype), heads, Nkv, [ops.PART_K, ops.PART_V], 1.0)
    # layouts
    qr = (x @ wq.T).reshape(B, Nq, heads, 64).transpose(0, 2, 1, 3) * scale
    if not fast:
        assert rel_err(n(q), qr) < 2e-5
        assert rel_err(n(k)[:, :, :Nkv], (c @ wk.T).reshape(B, Nkv, heads, 64).transpose(0, 2, 1, 3)) < 2e-5
        assert rel_err(n(vt)[:, :, :, :Nkv], (c @ wv.T).reshape(B, Nkv, heads, 64).transpose(0, 2, 3, 1)) < 2e-5
    out = n(ops.attention(q, k, vt, Nkv, use_exp2=fast))
    ref = _attention_ref((x @ wq.T).reshape(B, Nq, inner), (c @ wk.T).reshape(B, Nkv, inner), (c @ wv.T).reshape(B, Nkv, inner), heads, scale)
    assert rel_err(out, ref) < (3e-5 if dtype == torch.float32 else 3e-2), rel_err(out, ref)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("pattern", ["rising", "falling", "spikes", "huge_jumps", "very_negative"])
@pytest.mark.parametrize("Nkv", [416, 448])
def test_attention_running_max_rescale_paths(dtype, pattern, Nkv):
    """Scores whose maximum keeps growing along the key axis, including jumps far beyond the exp2 range (the bf16 kernel's
    fixed-reference fast path overflows and the workgroup must notice and take its exact path), shrinking, with isolated
    spikes, and all very negative.  Nkv = 416: 6.5 tiles, ragged last tile, odd number of half-tiles (fast steps, then exact
    steps); 448: whole tiles only (every half-tile through the fast step)."""
    B, H, Nq = 1, 2, 256
    rng = np.random.default_rng(7)
    q = rng.standard_normal((B, H, Nq, 64)).astype(np.float32)
    k = rng.standard_normal((B, H, Nkv, 64)).astype(np.float32) * 0.3
    v = rng.standard_normal((B, H, Nkv, 64)).astype(np.float32)
    qdir = q / np.linalg.norm(q, axis=-1, keepdims=True)
    ramp = np.linspace(0.0, 1.0, Nkv, dtype=np.float32)
    if pattern == "rising":
        boost = 60.0 * ramp                               # score of key j grows steadily: max moves at almost every tile
    elif pattern == "falling":
        boost = 60.0 * (1 - ramp)
    elif pattern == "spikes":
        boost = np.where(rng.random(Nkv) < 0.03, 45.0, 0.0).astype(np.float32)
    elif pattern == "huge_jumps":
        boost = (np.arange(Nkv) // 50).astype(np.float32) * 1500.0     # score jumps of several hundred (exp2 domain) every 50 keys
    else:
        boost = np.full(Nkv, -400.0, dtype=np.float32)
    # add boost_j along the mean query direction of each head: s_ij gains ~ boost_j * |q_i| * cos(...)
    mean_dir = qdir.mean(axis=2, keepdims=True)
    mean_dir /= np.linalg.norm(mean_dir, axis=-1, keepdims=True)
    k = k + boost[None, None, :, None] * mean_dir
    scale = 0.125
    fast = dtype == torch.bfloat16
    if fast:                                              # the kernel sees q pre-scaled by scale*log2(e), rounded ONCE
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
    nkp = 448
    kp = np.zeros((B, H, nkp, 64), np.float32)
    kp[:, :, :Nkv] = k
    vtp = np.zeros((B, H, 64, nkp), np.float32)
    vtp[:, :, :, :Nkv] = v.transpose(0, 1, 3, 2)
    if fast:
        ops.attention_fallbacks(reset=True)
    out = n(ops.attention(t(qs, dtype), t(kp, dtype), t(vtp, dtype), Nkv, use_exp2=fast))
    assert np.isfinite(out).all()
    # scores of ~1e4 carry an fp32 ulp of ~1e-3 themselves: any fp32 softmax is only that accurate against float64
    tol32 = 2e-3 if pattern == "huge_jumps" else 5e-5
    assert rel_err(out, ref) < (tol32 if dtype == torch.float32 else 4e-2), rel_err(out, ref)
    if fast:
        # the exact path ran where, and only where, the scores leave the fast path's range: huge_jumps climbs by hundreds of
        # octaves per 50 keys in every workgroup (2 heads x 4 blocks of 64 queries at this size); the other patterns stay within
        # 2^64 of the first keys' maximum
        fb = ops.attention_fallbacks(reset=True)
        assert (fb == 8) if pattern == "huge_jumps" else (fb == 0), fb


@pytest.mark.parametrize("dominance", [0.0, 10.0, 20.0, 45.0])
def test_attention_on_trained_like_peaky_heads_reports_its_fallback_rate(dominance):
    """What random-init models never show: heads with a logit scale of 8-12 nats (std of the scores) and a few
    dominant keys placed AFTER key 32, i.e. outside the prefix the bf16 fast path takes its reference maximum from, scoring
    `dominance` nats (+- 12 % by query) above a typical key.  The result must match a float64 softmax whatever path ran.  The
    fast path holds while l = sum_j 2^(s_j - m_prefix) < 2^64 (44 nats over the prefix maximum): the test derives from the
    float64 scores which 16-query tiles must / must not overflow, brackets the counter with that, prints the rate, and bounds
    the cost of a launch in which workgroups fall back to 4x the fast launch."""
    B, H, N = 2, 8, 1024
    rng = np.random.default_rng(11)
    sigma = np.linspace(8.0, 12.0, H, dtype=np.float32)                   # per-head logit scale, nats
    u = rng.standard_normal(64).astype(np.float32)
    u /= np.linalg.norm(u)
    q = rng.standard_normal((B, H, N, 64)).astype(np.float32)
    q += (8.0 - q @ u)[..., None] * u * 0.0 + 8.0 * u                     # a shared query component: q.u = 8 + N(0, 1)
    k = rng.standard_normal((B, H, N, 64)).astype(np.float32)
    k -= (k @ u)[..., None] * u                                           # ordinary keys ignore it ...
    k *= sigma[None, :, None, None]                                       # ... and score 0.125 * q.k ~ N(0, sigma_h)
    v = rng.standard_normal((B, H, N, 64)).astype(np.float32)
    for pos in (40, 333, 700, 1001):                                      # ... the dominant ones gain 0.125 * (8 + n) * c = dominance * (1 + n / 8)
        k[:, :, pos] += dominance * u
    k, v = bf16_round(k), bf16_round(v)
    qs = bf16_round(q * (0.125 * ops.LOG2E))
    s2 = qs.astype(np.float64) @ k.astype(np.float64).transpose(0, 1, 3, 2)          # scores in octaves, as the kernel sees them
    log2_l = np.log2(np.exp2(s2 - s2[..., :32].max(-1, keepdims=True)).sum(-1))       # per query; the kernel tests l < 2^64
    s = s2 * np.log(2.0)
    s -= s.max(-1, keepdims=True)
    pr = np.exp(s)
    pr /= pr.sum(-1, keepdims=True)
    ref = (pr @ v.astype(np.float64)).transpose(0, 2, 1, 3).reshape(B * N, H * 64)
    bf = torch.bfloat16
    tq, tk, tvt = t(qs, bf), t(k, bf), t(np.ascontiguousarray(v.transpose(0, 1, 3, 2)), bf)
    ops.attention_fallbacks(reset=True)
    out = n(ops.attention(tq, tk, tvt, N, use_exp2=True))
    fb = ops.attention_fallbacks(reset=True)
    must = (log2_l > 64.5).reshape(B, H, N // 256, 256).any(-1).sum()             # a workgroup holds at most 256 queries ...
    may = (log2_l > 63.5).reshape(B, H, N // 16, 16).any(-1).sum()                # ... and decides per 16-query tile

    def timed(a, b, c):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ops.attention(a, b, c, N, use_exp2=True)
        e0.record()
        for _ in range(10):
            ops.attention(a, b, c, N, use_exp2=True)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10

    ms = timed(tq, tk, tvt)
    g = torch.Generator().manual_seed(1)
    plain = [(torch.randn(B, H, N, 64, generator=g) * 0.18).to(dev()).to(bf), torch.randn(B, H, N, 64, generator=g).to(dev()).to(bf),
             torch.randn(B, H, 64, N, generator=g).to(dev()).to(bf)]
    ms_plain = timed(*plain)
    ops.attention_fallbacks(reset=True)
    print(f"dominance {dominance} nats, head logit scale 8-12 nats: max l over the 32-key prefix reference 2^{float(log2_l.max()):.1f} "
          f"(limit 2^64), workgroups re-run {fb} (must {int(must)}, may {int(may)}), {ms * 1e3:.1f} us per launch vs {ms_plain * 1e3:.1f} us "
          f"on random data")
    assert np.isfinite(out).all() and rel_err(out, ref) < 4e-2, rel_err(out, ref)
    assert (fb >= 1 if must else True) and fb <= may, (fb, int(must), int(may))
    assert ms < 4.0 * ms_plain + 0.05, (ms, ms_plain)


def test_attention_fallback_is_per_workgroup_and_matches_the_exact_result():
    """One (batch, head) of eight carries a key that overflows the fast path: exactly its query blocks are redone, the other
    heads are untouched bit for bit, and the redone head equals what the exact path gives on its own (Nkv = 100 is ragged and
    below three tiles: exact path only)."""
    B, H, N = 2, 4, 512
    rng = np.random.default_rng(11)
    q = bf16_round(rng.standard_normal((B, H, N, 64)).astype(np.float32) * 0.3)
    k = bf16_round(rng.standard_normal((B, H, N, 64)).astype(np.float32))
    v = bf16_round(rng.standard_normal((B, H, N, 64)).astype(np.float32))
    vt = np.ascontiguousarray(v.transpose(0, 1, 3, 2))
    ops.attention_fallbacks(reset=True)
    base = n(ops.attention(t(q, torch.bfloat16), t(k, torch.bfloat16), t(vt, torch.bfloat16), N, use_exp2=True))
    assert ops.attention_fallbacks(reset=True) == 0
    k2 = k.copy()
    k2[1, 2, 300] = bf16_round(q[1, 2, 5] * 4000.0)       # a key aligned with one query: ~ +1e4 octaves for that row, large for the others
    out = n(ops.attention(t(q, torch.bfloat16), t(k2, torch.bfloat16), t(vt, torch.bfloat16), N, use_exp2=True))
    fb = ops.attention_fallbacks(reset=True)
    assert 1 <= fb <= N // 64, fb                           # only query blocks of (b=1, h=2) (64 queries each at this size)
    assert np.isfinite(out).all()
    o4, b4 = out.reshape(B, N, H, 64), base.reshape(B, N, H, 64)
    mask = np.ones((B, H), bool)
    mask[1, 2] = False
    assert all(np.array_equal(o4[b, :, h], b4[b, :, h]) for b in range(B) for h in range(H) if mask[b, h])
    s = (q[1, 2].astype(np.float64) @ k2[1, 2].astype(np.float64).T) * np.log(2.0)
    s -= s.max(-1, keepdims=True)
    pr = np.exp(s)
    pr /= pr.sum(-1, keepdims=True)
    assert rel_err(o4[1, :, 2], pr @ v[1, 2].astype(np.float64)) < 4e-2


@pytest.mark.parametrize("Nkv", [512, 416, 77])
def test_attention_result_does_not_depend_on_the_batch_or_the_workgroup_size(Nkv):
    """The bf16 kernel runs 256 / 128 / 64 queries per workgroup depending on how many workgroups the launch has (small batches
    would leave most CUs idle).  An image's rows must come out bit-identical whichever size serves it -- including the 16-query
    tile that overflows the fast path and its neighbours (the fallback is decided per tile, never per workgroup)."""
    B, H, N = 32, 8, 512
    Np = (Nkv + 63) // 64 * 64                                  # 512: whole tiles, every half-tile through the fast step; 416: fast steps, then
    rng = np.random.default_rng(5)                              # exact steps over the ragged tail; 77 (cross-attention): exact path only
    q = bf16_round(rng.standard_normal((B, H, N, 64)).astype(np.float32) * 0.3)
    k = bf16_round(rng.standard_normal((B, H, Np, 64)).astype(np.float32))
    vt = bf16_round(rng.standard_normal((B, H, 64, Np)).astype(np.float32))
    k[0, 3, min(200, Nkv - 3)] = bf16_round(q[0, 3, 77] * 4000.0)   # one head of image 0 overflows for the queries aligned with q[77]
    run = lambda sl: n(ops.attention(t(q[sl], torch.bfloat16), t(k[sl], torch.bfloat16), t(vt[sl], torch.bfloat16), Nkv, use_exp2=True))
    fast = Nkv > 128                                            # (a context of one or two tiles never enters the fast path: nothing can overflow)
    ops.attention_fallbacks(reset=True)
    full = run(slice(0, B)).reshape(B, N, H * 64)               # 32 * 8 * 2 = 512 workgroups of 256 queries
    assert (ops.attention_fallbacks(reset=True) >= 1) == fast and np.isfinite(full).all()
    for nb in (16, 3, 1):                                       # 128 queries per workgroup; 64; 64
        part = run(slice(0, nb)).reshape(nb, N, H * 64)
        assert np.array_equal(part, full[:nb]), nb
    assert (ops.attention_fallbacks(reset=True) >= 3) == fast
    tail = run(slice(B - 2, B)).reshape(2, N, H * 64)           # images without an overflow
    assert np.array_equal(tail, full[B - 2:]) and ops.attention_fallbacks(reset=True) == 0


def test_attention_ignores_garbage_in_padding():
    """K rows / V^T columns beyond Nkv may hold anything (NaN included)."""
    B, H, Nq, Nkv = 1, 2, 64, 77
    q, k, v = rnd(B, H, Nq, 64), rnd(B, H, 128, 64), rnd(B, H, 128, 64)
    k2, v2 = k.copy(), v.copy()
    k2[:, :, Nkv:] = np.nan
    v2[:, :, Nkv:] = np.inf
    a = n(ops.attention(t(q), t(k), t(np.ascontiguousarray(v.transpose(0, 1, 3, 2))), Nkv))
    b = n(ops.attention(t(q), t(k2), t(np.ascontiguousarray(v2.transpose(0, 1, 3, 2))), Nkv))
    assert np.array_equal(a, b) and np.isfinite(a).all()


@pytest.mark.parametrize("D", [64, 128, 512, 768, 1024, 1280])
@pytest.mark.parametrize("out_dtype", [torch.float32, torch.bfloat16])
def test_layernorm(D, out_dtype):
    x, g, b = rnd(300, D, scale=3.0) + 0.5, rnd(D) + 1, rnd(D)
    ref = O.layernorm(x, g, b)
    out = n(ops.layernorm(t(x), t(g), t(b), 1e-5, out_dtype))
    assert rel_err(out, ref) < (1e-5 if out_dtype == torch.float32 else 1e-2)


@pytest.mark.parametrize("size,patch", [(32, 8), (256, 8), (64, 16)])
def test_patchify_unpatchify_exact(size, patch):
    img = rnd(2, 3, size, size)
    assert np.array_equal(n(ops.patchify(t(img), patch)), O.patchify(img, patch).reshape(-1, 3 * patch * patch))
    y = rnd(2 * (size // patch) ** 2, patch * patch * 3, scale=2.0)
    ref = np.clip(O.unpatchify(y.reshape(2, -1, patch * patch * 3), 3, size, patch), -1, 1)
    assert np.array_equal(n(ops.unpatchify_clamp(t(y), 2, 3, size, patch)), ref)


def test_row_utilities_exact():
    x = rnd(100, 32)
    out = n(ops.convert_pad(t(x), 64, torch.float32))
    assert np.array_equal(out[:, :32], x) and np.all(out[:, 32:] == 0)
    outb = n(ops.convert_pad(t(x), 64, torch.bfloat16))
    assert np.array_equal(outb[:, :32], bf16_round(x))
    table, ids = rnd(65, 32), RNG.integers(0, 65, 500)
    g = n(ops.embed_rows(t(table), t(ids.astype(np.int64)), 64, torch.float32))
    assert np.array_equal(g[:, :32], table[ids]) and np.all(g[:, 32:] == 0)
    xx, pos = rnd(48, 64), rnd(16, 64)
    assert np.array_equal(n(ops.add_rows(t(xx), t(pos))), xx + pos[np.arange(48) % 16])


@pytest.mark.parametrize("M,V,E", [(48, 64, 32), (3000, 8192, 32), (1024, 1000, 16)])
def test_vq_bit_exact_against_c_oracle(M, V, E):
    cb, z = rnd(V, E), rnd(M, E, scale=0.3)
    en_r, sq_r = vq_ref.prepare(cb)
    idx_r, zn_r, dmin, gap = vq_ref.quantize(z, en_r, sq_r)
    en, sq = ops.vq_prepare(t(cb))
    assert np.array_equal(n(en), en_r) and np.array_equal(n(sq), sq_r)           # bit-exact preparation
    z_out, idx, loss = ops.vq_quantize(t(z), en, sq, 0.25)
    assert np.array_equal(n(idx), idx_r)                                          # bit-exact indices
    zq = en_r[idx_r]
    assert np.array_equal(n(z_out), zn_r + (zq - zn_r))                           # z + (z_q - z), quantize.py:36
    m = np.mean((zq - zn_r) ** 2, dtype=np.float64)
    assert abs(float(n(loss)[0]) - 1.25 * m) < 1e-6 * max(1.0, m)
    # and against the numpy restatement of the reference formula
    _, loss_o, idx_o = O.vq_forward(z, cb)
    agree = np.mean(idx_o.reshape(-1) == idx_r)
    assert agree == 1.0 or np.all(gap[idx_o.reshape(-1) != idx_r] < 1e-6)


@pytest.mark.parametrize("V,topk,temp", [(64, 1, 1.0), (64, 5, 0.7), (8192, 5, 1.0), (8192, 1, 0.0), (1000, 16, 0.3), (8192, 64, 2.0)])
def test_sample_rows_with_given_noise(V, topk, temp):
    M = 96
    logits = rnd(M, V, scale=2.0)
    ids = RNG.integers(0, V, M).astype(np.int64)
    ids[RNG.random(M) < 0.6] = V
    noise = RNG.random((M, V)).astype(np.float32)
    pred_r, merged_r, score_r = O.sample_rows(logits, ids, V, topk, temp, noise)
    pred, merged, score = ops.sample_rows(t(logits), t(ids), V, topk, temp, noise=t(noise))
    assert np.array_equal(n(pred), pred_r)
    assert np.array_equal(n(merged), merged_r)
    assert np.max(np.abs(n(score) - score_r)) < 2e-6


@pytest.mark.parametrize("V,topk", [(64, 1), (64, 5), (128, 8), (256, 3), (1024, 5), (4096, 8), (8192, 5), (8192, 8), (12288, 2), (16384, 5)])
@pytest.mark.parametrize("ties", [False, True])
def test_sample_rows_from_block_statistics(V, topk, ties):
    """Round 5: for top-k <= 8 the sampling kernel works from the softmax statistics of the row's 64-column blocks (max, sum of exp)
    and reads only the k blocks with the largest maxima.  Handed in (pmhip_sample_rows_stats) or derived from the stored row by the
    same arithmetic (pmhip_sample_rows), the result is the same bit for bit, and it is the oracle's (ties included: quantised
    logits put equal values in different blocks, equal block maxima, and equal values inside one block)."""
    M = 70
    logits = rnd(M, V, scale=3.0)
    if ties:
        logits = (np.round(logits * 2) / 2).astype(np.float32)
    ids = RNG.integers(0, V, M).astype(np.int64)
    ids[RNG.random(M) < 0.6] = V
    noise = RNG.random((M, V)).astype(np.float32)
    pred_r, merged_r, score_r = O.sample_rows(logits, ids, V, topk, 0.8, noise)
    x = t(logits)
    same, stats = ops.guidance_combine(x, x, 1.0, with_stats=True)       # u + 1 * (c - u) with c = u: the logits themselves
    assert torch.equal(same, x) and stats.shape == (M, V // 64, 2)
    blocks = logits.reshape(M, V // 64, 64)
    assert np.array_equal(n(stats[..., 0]), blocks.max(-1))
    want_sum = np.exp(blocks.astype(np.float64) - blocks.max(-1, keepdims=True)).sum(-1)
    assert np.max(np.abs(n(stats[..., 1]) / want_sum - 1)) < 2e-6
    dense = ops.sample_rows(x, t(ids), V, topk, 0.8, noise=t(noise))
    sparse = ops.sample_rows(x, t(ids), V, topk, 0.8, noise=t(noise), block_stats=stats)
    for a, b in zip(dense, sparse):
        assert torch.equal(a, b)
    assert np.array_equal(n(dense[0]), pred_r) and np.array_equal(n(dense[1]), merged_r)
    assert np.max(np.abs(n(dense[2]) - score_r)) < 2e-6
    # Philox noise: same draws either way
    d2 = ops.sample_rows(x, t(ids), V, topk, 0.8, seed=77, step=3, row_base=1234)
    s2 = ops.sample_rows(x, t(ids), V, topk, 0.8, seed=77, step=3, row_base=1234, block_stats=stats)
    for a, b in zip(d2, s2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M", [1024, 8192])                    # the 128x128 kernel / the 256x256 kernel
@pytest.mark.parametrize("fold", [False, True])
def test_logits_gemm_leaves_the_block_statistics_of_what_it_stores(M, fold):
    """pmhip_gemm_softmax_stats: the logits are those of the plain call bit for bit; the statistics are the ones the sampling kernel
    derives from the stored logits (same arithmetic, common.h softmax_block_stat) bit for bit -- checked through the kernel that
    computes them from a stored plane (guidance_combine with cond = uncond)."""
    D, V = 512, 8192
    w, b = bf16_round(rnd(V, D, scale=D ** -0.5)), rnd(V)
    if fold:
        hi, _ = ops.split_hilo(t(rnd(M, D) + 0.4))
        coef = ops.ln_coef(hi)
        wg, c, d = packing.ln_fold(t(w), t(1 + 0.3 * rnd(D)), t(0.2 * rnd(D)))
        plain = ops.gemm_ln(hi, wg, coef, c, d, bias=t(b), out_dtype=torch.float32)
        logits, stats = ops.gemm_softmax_stats(hi, wg, bias=t(b), fold=(coef, c, d))
    else:
        a = t(bf16_round(rnd(M, D)), torch.bfloat16)
        plain = ops.gemm(a, t(w, torch.bfloat16), bias=t(b), out_dtype=torch.float32)
        logits, stats = ops.gemm_softmax_stats(a, t(w, torch.bfloat16), bias=t(b))
    assert torch.equal(logits, plain)
    _, want = ops.guidance_combine(logits, logits, 1.0, with_stats=True)
    assert torch.equal(stats, want)
    ids = torch.full((M,), V, device=dev(), dtype=torch.int64)
    for a_, b_ in zip(ops.sample_rows(logits, ids, V, 5, 0.7, seed=5, step=1), ops.sample_rows(logits, ids, V, 5, 0.7, seed=5, step=1, block_stats=stats)):
        assert torch.equal(a_, b_)


def test_sample_rows_philox_matches_numpy_philox_and_is_shard_invariant():
    M, V, topk = 64, 8192, 5
    logits = rnd(M, V, scale=2.0)
    ids = np.full(M, V, dtype=np.int64)
    seed, step, base = 0x1234567890ABCDEF, 3, 7000
    cols = np.broadcast_to(np.arange(V), (M, V))
    rows = np.broadcast_to((base + np.arange(M))[:, None], (M, V))
    noise = O.philox_uniform(seed, step, rows, cols)
    pred_r, _, score_r = O.sample_rows(logits, ids, V, topk, 0.9, noise)
    pred, _, score = ops.sample_rows(t(logits), t(ids), V, topk, 0.9, seed=seed, step=step, row_base=base)
    assert np.array_equal(n(pred), pred_r)
    # the second half computed alone with the matching row_base gives the same draws
    pred_b, _, _ = ops.sample_rows(t(logits[32:]), t(ids[32:]), V, topk, 0.9, seed=seed, step=step, row_base=base + 32)
    assert np.array_equal(n(pred_b), pred_r[32:])
    pred_c, _, _ = ops.sample_rows(t(logits), t(ids), V, topk, 0.9, seed=seed, step=step + 1, row_base=base)
    assert not np.array_equal(n(pred_c), pred_r)


@pytest.mark.parametrize("B,N,m", [(3, 16, 8), (4, 1024, 1004), (4, 1024, 1), (2, 1024, 391), (2, 100, 37), (2, 4096, 2000), (1, 2048, 5),
                                   (2, 513, 100), (3, 257, 256), (1, 1, 1), (2, 300, 300), (33, 1024, 722)])
def test_remask_exact_with_ties(B, N, m):
    scores = np.round(RNG.random((B, N)).astype(np.float32), 2)        # heavy ties on purpose
    scores[:, ::7] = -1e5
    ids = RNG.integers(0, 50, (B, N)).astype(np.int64)
    ref = O.remask(ids, scores, m, 8192)
    out = n(ops.remask(t(ids), t(scores), m, 8192))
    assert np.array_equal(out, ref)
    assert np.all((out == 8192).sum(1) == m)


# ---- masked-token objective kernels (loss.hip) -------------------------------------------------------
@pytest.mark.parametrize("B,N,E,ratio", [(2, 16, 8, 0.75), (3, 1024, 32, 0.55), (1, 1500, 32, 0.999), (2, 64, 4, 0.0)])
def test_random_mask_matches_oracle_with_ties(B, N, E, ratio):
    rng = np.random.default_rng(B * N + E)
    z = rng.standard_normal((B, N, E)).astype(np.float32)
    noise = (rng.integers(0, max(N // 3, 2), (B, N)) / N).astype(np.float32)      # many ties: stable order decides
    tok = rng.standard_normal(E).astype(np.float32)
    len_keep = N - max(int(N * ratio), 1)
    x, mask = ops.random_mask(t(z), t(noise), t(tok), len_keep)
    xo, mo = O.random_masking(z, tok, ratio, noise)
    assert np.array_equal(n(mask), mo) and np.array_equal(n(x), xo)
    assert n(mask).sum(1).tolist() == [N - len_keep] * B


@pytest.mark.parametrize("M,V,eps", [(48, 64, 0.1), (300, 1000, 0.0), (4096, 8192, 0.1), (17, 8200, 0.3)])
def test_masked_ce_matches_oracle(M, V, eps):
    rng = np.random.default_rng(M + V)
    logits = (rng.standard_normal((M, V)) * 4).astype(np.float32)
    labels = rng.integers(0, V, M)
    mask = (rng.random(M) < 0.5).astype(np.float32)
    loss, rows = ops.masked_ce(t(logits), t(labels), t(mask), eps)
    lo, ro = O.masked_ce(logits, labels, mask, eps)
    assert maxabs(n(rows), ro) < 1e-4
    assert abs(float(loss) - float(lo)) < 1e-4
    ref = torch.nn.functional.cross_entropy(torch.from_numpy(logits), torch.from_numpy(labels), label_smoothing=eps,
                                            reduction="none").numpy()
    assert maxabs(n(rows), ref * mask) < 1e-4


def test_c_abi_rejects_bad_arguments_with_a_message_and_stays_usable():
    """Error behaviour of the boundary (include/pmhip.h): a bad call returns a non-zero code, pmhip_last_error() names the
    problem, nothing is launched, and the library keeps working afterwards -- the Python shim turns the code into PmhipError
    (a RuntimeError, as the reference raises from torch for the same mistakes)."""
    bf = torch.bfloat16
    a = torch.zeros(64, 100, device=dev(), dtype=bf)               # K = 100: not a multiple of 64
    w = torch.zeros(64, 100, device=dev(), dtype=bf)
    with pytest.raises(PmhipError, match="multiple of 64"):
        ops.gemm(a, w)
    q = torch.zeros(1, 2, 64, 64, device=dev(), dtype=bf)
    k = torch.zeros(1, 2, 64, 64, device=dev(), dtype=bf)
    vt = torch.zeros(1, 2, 64, 64, device=dev(), dtype=bf)
    with pytest.raises(PmhipError, match="Nkv_pad"):
        ops.attention(q, k, vt, 100)                                # more keys than the padded K / V^T hold
    logits = torch.zeros(4, 64, device=dev())
    ids = torch.zeros(4, dtype=torch.int64, device=dev())
    with pytest.raises(PmhipError, match="topk"):
        ops.sample_rows(logits, ids, 64, 65, 1.0, noise=torch.full((4, 64), 0.5, device=dev()))
    lib = _lib.load()
    assert lib.pmhip_attention(1, None, None, None, None, 64, 1, 1, 64, 64, 64, 1, None) != 0
    assert b"null" in lib.pmhip_last_error()
    assert lib.pmhip_timing_get(b"no-such-family", None, None) != 0
    # ... and the next good call is unaffected
    a2 = t(bf16_round(rnd(128, 128)), bf)
    w2 = t(bf16_round(rnd(128, 128)), bf)
    assert rel_err(n(ops.gemm(a2, w2, out_dtype=torch.float32)), n(a2.float()) @ n(w2.float()).T) < 1e-5


def test_masked_ce_nothing_masked_is_nan_and_bad_args_raise():
    logits = torch.zeros(8, 64, device=dev())
    labels = torch.zeros(8, dtype=torch.int64, device=dev())
    loss, _ = ops.masked_ce(logits, labels, torch.zeros(8, device=dev()), 0.1)
    assert np.isnan(float(loss))
    with pytest.raises(PmhipError):
        ops.masked_ce(logits, labels, torch.zeros(8, device=dev()), 1.5)
    with pytest.raises(PmhipError):
        ops.random_mask(torch.zeros(1, 8, 4, device=dev()), torch.zeros(1, 8, device=dev()), torch.zeros(4, device=dev()), 9)


@pytest.mark.parametrize("M,N,K", [(19200, 1024, 128), (65792, 256, 192), (8192, 3072, 320)])
@pytest.mark.parametrize("out_dtype", [torch.bfloat16, torch.float32])
def test_gemm_large_tile_kernel_streamed_tiles(M, N, K, out_dtype):
    """The persistent 256x256 kernel streams the K-tiles of consecutive output tiles as one DMA sequence: tile counts that do
    not divide by the number of workgroups (300, 257 = one workgroup with two tiles, 384) and short K loops (2, 3 and 5 K-tiles:
    first + last only, one steady tile, an odd count so that the buffer parity flips from tile to tile)."""
    a, w, b = bf16_round(rnd(M, K)), bf16_round(rnd(N, K, scale=K ** -0.5)), rnd(N)
    ref = a @ w.T + b
    for _ in range(2):                                   # twice: the second launch finds warm caches / another timing
        out = n(ops.gemm(t(a, torch.bfloat16), t(w, torch.bfloat16), bias=t(b), out_dtype=out_dtype))
        assert rel_err(out, ref) < (2e-5 if out_dtype == torch.float32 else 1e-2), rel_err(out, ref)
        blk = np.abs(out - ref).reshape(M // 256, 256, N // 256, 256).max(axis=(1, 3))
        assert blk.max() < (1e-3 if out_dtype == torch.float32 else 0.15 * np.abs(ref).max())


def test_gemm_large_tile_kernel_streamed_residual_and_ragged_heads():
    """(a) residual f32 GEMM on the 256x256 kernel with 150 tiles (K = 1024); (b) head split with 336 tokens per image (padded to 384): the
    wave's 128 rows straddle images, so the Q / K epilogue takes its per-row (batch, token) path and V its scalar path."""
    M, N, K = 19200, 512, 1024
    a, w, b, r = bf16_round(rnd(M, K)), bf16_round(rnd(N, K, scale=K ** -0.5)), rnd(N), rnd(M, N)
    out = n(ops.gemm(t(a, torch.bfloat16), t(w, torch.bfloat16), bias=t(b), residual=t(r), out_dtype=torch.float32))
    assert rel_err(out, a @ w.T + b + r) < 2e-5
    B, heads, N_tok, D = 64, 8, 336, 512
    x = bf16_round(rnd(B * N_tok, D))
    wqkv = bf16_round(rnd(3 * heads * 64, D, scale=D ** -0.5))
    q, k, vt = ops.gemm_heads(t(x, torch.bfloat16), t(wqkv, torch.bfloat16), heads, N_tok, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.125)
    full = (x @ wqkv.T).reshape(B, N_tok, 3, heads, 64)
    assert rel_err(n(q), full[:, :, 0].transpose(0, 2, 1, 3) * 0.125) < 1e-2
    assert rel_err(n(k)[:, :, :N_tok], full[:, :, 1].transpose(0, 2, 1, 3)) < 1e-2
    assert rel_err(n(vt)[:, :, :, :N_tok], full[:, :, 2].transpose(0, 2, 3, 1)) < 1e-2
    assert float(k[:, :, N_tok:].abs().max()) == 0 and float(vt[:, :, :, N_tok:].abs().max()) == 0


def test_streamed_kernels_are_bit_stable_over_repetitions():
    """Race screen for the hand-placed waits (counted vmcnt / lgkmcnt, inline-asm LDS reads) of gemm256 and attention: 150
    launches each at the bench shapes, other kernels in between, must all produce the bits of the first launch."""
    g = torch.Generator().manual_seed(7)
    M, D = 65536, 512
    a = (torch.randn(M, D, generator=g) * 0.5).to(torch.bfloat16).to(dev())
    wq = (torch.randn(1536, D, generator=g) * D ** -0.5).to(torch.bfloat16).to(dev())
    w12 = (torch.randn(2816, D, generator=g) * D ** -0.5).to(torch.bfloat16).to(dev())
    b12 = torch.randn(2816, generator=g).to(dev())
    wl = (torch.randn(2048, D, generator=g) * D ** -0.5).to(torch.bfloat16).to(dev())
    q0, k0, v0 = ops.gemm_heads(a, wq, 8, 1024, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.125 * ops.LOG2E)
    h0 = ops.gemm_swiglu(a, w12, b12)
    l0 = ops.gemm(a, wl, out_dtype=torch.float32)
    o0 = ops.attention(q0, k0, v0, 1024, use_exp2=True)
    for i in range(150):
        q, k, v = ops.gemm_heads(a, wq, 8, 1024, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.125 * ops.LOG2E)
        o = ops.attention(q, k, v, 1024, use_exp2=True)
        h = ops.gemm_swiglu(a, w12, b12)
        l = ops.gemm(a, wl, out_dtype=torch.float32)
        assert torch.equal(q, q0) and torch.equal(k, k0) and torch.equal(v, v0), i
        assert torch.equal(o, o0), i
        assert torch.equal(h, h0) and torch.equal(l, l0), i


def test_guidance_combine_is_one_fma_per_element():
    """pmhip_guidance_combine: out = fmaf(scale, cond - uncond, uncond) in fp32, in place allowed, scale 0 / 1 edge cases"""
    g = torch.Generator().manual_seed(3)
    c = torch.randn(5, 1024, 512, generator=g).to(dev()) * 4
    u = torch.randn(5, 1024, 512, generator=g).to(dev()) * 4
    for scale in (0.0, 1.0, 2.5, -0.75):
        out = ops.guidance_combine(c, u, scale)
        diff = (c - u).double()                              # the fp32 difference, exactly representable in float64
        want = (diff * float(np.float32(scale)) + u.double()).float()     # one rounding of the exact fused result
        assert torch.equal(out, want), scale
    assert torch.equal(ops.guidance_combine(c, u, 0.0), u)
    c2 = c.clone()
    assert ops.guidance_combine(c2, u, 2.5, out=c2) is c2 and torch.equal(c2, ops.guidance_combine(c, u, 2.5))
    with pytest.raises(ValueError):
        ops.guidance_combine(c, u[:4], 1.0)
