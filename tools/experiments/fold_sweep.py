import sys, os
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import torch, gemm_bench as g
from paintmind_amd import ops, packing
dev = torch.device("cuda:0")
M, D = 65536, 512
x = torch.randn(M, D, device=dev)
a = (torch.randn(M, D, device=dev) * 0.5).to(torch.bfloat16)
wo = (torch.randn(D, D, device=dev) * D ** -0.5).to(torch.bfloat16)
bo = torch.randn(D, device=dev)
gamma = torch.rand(D, device=dev) + 0.5; beta = torch.randn(D, device=dev)
out, xb, stats = ops.gemm_stats(a, wo, bias=bo, residual=x)
y = ops.layernorm(out, gamma, beta, out_dtype=torch.bfloat16)
for N in (512, 1024, 1536, 2048, 2816, 4096):
    wq = (torch.randn(N, D, device=dev) * D ** -0.5)
    wg, c, d = packing.ln_fold(wq, gamma, beta, torch.bfloat16)
    wqb = wq.to(torch.bfloat16)
    t0 = g.timeit(lambda: ops.gemm(y, wqb, out_dtype=torch.bfloat16)) * 1e3
    t1 = g.timeit(lambda: ops.gemm_ln(xb, wg, stats, c, d, out_dtype=torch.bfloat16)) * 1e3
    t2 = g.timeit(lambda: ops.gemm(xb, wqb, out_dtype=torch.bfloat16)) * 1e3
    print(f"N={N}: plain(y) {t0:.1f}  fold {t1:.1f}  plain(xb) {t2:.1f}")
