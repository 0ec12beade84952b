"""Do HBM-bound kernels (LayerNorm) overlap MFMA-bound ones (gemm256 variants, attention) launched on ANOTHER stream?
A LayerNorm wave needs 40 VGPRs; two gemm256 waves per SIMD leave 512 - 2 * alloc(VGPRs) registers free.
    python tools/overlap_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paintmind_amd import ops
dev = torch.device("cuda:0")
M, D = 65536, 512
a = (torch.rand(M, D, device=dev) * 2 - 1).to(torch.bfloat16)
x = torch.rand(M, D, device=dev); gm = torch.rand(D, device=dev); bt = torch.rand(D, device=dev)
w12 = (torch.rand(2816, D, device=dev) * 2 - 1).to(torch.bfloat16); b12 = torch.rand(2816, device=dev)
wq = (torch.rand(1536, D, device=dev) * 2 - 1).to(torch.bfloat16)
wl = (torch.rand(8192, D, device=dev) * 2 - 1).to(torch.bfloat16)
q = ((torch.rand(64, 8, 1024, 64, device=dev) * 2 - 1) * 0.5).to(torch.bfloat16); k = torch.rand_like(q.float()).to(torch.bfloat16); vt = torch.rand(64, 8, 64, 1024, device=dev).to(torch.bfloat16)
heavy = {
    "gemm256 SwiGLU (225 VGPRs)": (lambda: ops.gemm_swiglu(a, w12, b12), 40),
    "gemm256 head-split QKV (255)": (lambda: ops.gemm_heads(a, wq, 8, 1024, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.125), 60),
    "gemm256 logits f32 (226)": (lambda: ops.gemm(a, wl, out_dtype=torch.float32), 12),
    "attention (250, 2 WGs/CU)": (lambda: ops.attention(q, k, vt, 1024, use_exp2=True), 50),
}
ln = lambda: ops.layernorm(x, gm, bt, out_dtype=torch.bfloat16)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fa, na, fb, nb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fa:
        with torch.cuda.stream(sa):
            for _ in range(na): fa()
    if fb:
        with torch.cuda.stream(sb):
            for _ in range(nb): fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3
for name, (f, n) in heavy.items():
    nl = 200
    for _ in range(2): timed(f, n, ln, nl)
    ta, tb, tab = min(timed(f, n, None, 0) for _ in range(3)), min(timed(None, 0, ln, nl) for _ in range(3)), min(timed(f, n, ln, nl) for _ in range(3))
    print(f"{name:32s} alone {ta:6.2f} ms | {nl} LayerNorms alone {tb:5.2f} ms | together {tab:6.2f} ms  (sum {ta+tb:6.2f}, overlap {100*(ta+tb-tab)/min(ta,tb):4.0f} % of the shorter)")
