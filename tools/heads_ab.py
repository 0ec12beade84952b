"""interleaved A/B: head-split projection plain (on LN output y, and on the raw hi plane) vs LayerNorm-folded"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, gemm_bench as g
from paintmind_amd import ops, packing
dev = torch.device("cuda:0")
M, D = 65536, 512
bf = torch.bfloat16
x = torch.randn(M, D, device=dev)
gamma = torch.rand(D, device=dev) + 0.5; beta = torch.randn(D, device=dev)
hi, lo = ops.split_hilo(x)
coef = ops.ln_coef(hi)
y = ops.layernorm_hilo(hi, lo, gamma, beta, out_dtype=bf)
for N in (1536, 512):
    wq = (torch.randn(N, D, device=dev) * D ** -0.5)
    wg, c, d = packing.ln_fold(wq, gamma, beta, bf)
    wqb = wq.to(bf)
    kinds = [ops.PART_Q, ops.PART_K, ops.PART_V][: N // 512]
    fns = {"plain(y)": lambda: ops.gemm_heads(y, wqb, 8, 1024, kinds, 0.125),
           "plain(hi)": lambda: ops.gemm_heads(hi, wqb, 8, 1024, kinds, 0.125),
           "plain(hi, wg)": lambda: ops.gemm_heads(hi, wg, 8, 1024, kinds, 0.125),
           "folded": lambda: ops.gemm_heads_ln(hi, wg, 8, 1024, kinds, 0.125, coef, c, d)}
    res = {k: [] for k in fns}
    for r in range(4):
        for k, f in fns.items():
            res[k].append(g.timeit(f) * 1e3)
    print("N =", N, {k: [round(v) for v in vs] for k, vs in res.items()})
