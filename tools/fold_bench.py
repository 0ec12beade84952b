"""bf16 hi/lo residual stream + LayerNorm fold: producer / coefficient / consumer kernels against the fp32-stream kernels at
the bench shapes (python tools/fold_bench.py)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, gemm_bench as g
from paintmind_amd import ops, packing
dev = torch.device("cuda:0")
M, D = 65536, 512
bf = torch.bfloat16
x = torch.randn(M, D, device=dev)
a = (torch.randn(M, D, device=dev) * 0.5).to(bf)
wo = (torch.randn(D, D, device=dev) * D ** -0.5).to(bf)
bo = torch.randn(D, device=dev)
gamma = torch.rand(D, device=dev) + 0.5; beta = torch.randn(D, device=dev)
hi, lo = ops.split_hilo(x)
print("producer fp32 stream (K=512) :", g.timeit(lambda: ops.gemm(a, wo, bias=bo, residual=x, out_dtype=torch.float32)) * 1e3, "us")
print("producer hi/lo stream        :", g.timeit(lambda: ops.gemm_hilo(a, wo, hi, lo, bias=bo)) * 1e3, "us")
hid = (torch.randn(M, 1408, device=dev) * 0.5).to(bf); w3 = (torch.randn(D, 1408, device=dev) * 1408 ** -0.5).to(bf)
print("producer fp32 stream (K=1408):", g.timeit(lambda: ops.gemm(hid, w3, bias=bo, residual=x, out_dtype=torch.float32)) * 1e3, "us")
print("producer hi/lo stream        :", g.timeit(lambda: ops.gemm_hilo(hid, w3, hi, lo, bias=bo)) * 1e3, "us")
print("layernorm (fp32 stream)      :", g.timeit(lambda: ops.layernorm(x, gamma, beta, out_dtype=bf)) * 1e3, "us")
print("layernorm_hilo (unfolded)    :", g.timeit(lambda: ops.layernorm_hilo(hi, lo, gamma, beta, out_dtype=bf)) * 1e3, "us")
print("ln_coef (hi plane only)      :", g.timeit(lambda: ops.ln_coef(hi)) * 1e3, "us")
coef = ops.ln_coef(hi)
y = ops.layernorm_hilo(hi, lo, gamma, beta, out_dtype=bf)
wq = (torch.randn(1536, D, device=dev) * D ** -0.5)
wg, c, d = packing.ln_fold(wq, gamma, beta, bf)
wqb = wq.to(bf)
print("heads plain     :", g.timeit(lambda: ops.gemm_heads(y, wqb, 8, 1024, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.125)) * 1e3, "us")
print("heads_ln        :", g.timeit(lambda: ops.gemm_heads_ln(hi, wg, 8, 1024, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.125, coef, c, d)) * 1e3, "us")
w12 = (torch.randn(2816, D, device=dev) * D ** -0.5); b12 = torch.randn(2816, device=dev)
w12g, c2, d2 = packing.ln_fold(w12, gamma, beta, bf)
print("swiglu plain    :", g.timeit(lambda: ops.gemm_swiglu(y, w12.to(bf), b12)) * 1e3, "us")
print("swiglu_ln       :", g.timeit(lambda: ops.gemm_swiglu_ln(hi, w12g, b12, coef, c2, d2)) * 1e3, "us")
wl = (torch.randn(8192, D, device=dev) * D ** -0.5); bl = torch.randn(8192, device=dev)
wlg, c3, d3 = packing.ln_fold(wl, gamma, beta, bf)
print("logits plain    :", g.timeit(lambda: ops.gemm(y, wl.to(bf), bias=bl, out_dtype=torch.float32)) * 1e3, "us")
print("logits_ln       :", g.timeit(lambda: ops.gemm_ln(hi, wlg, coef, c3, d3, bias=bl, out_dtype=torch.float32)) * 1e3, "us")
