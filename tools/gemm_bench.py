"""micro-benchmark of the GEMM / attention ops (development aid): python tools/gemm_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paintmind_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def gemm(M, N, K, out_dtype=torch.bfloat16, residual=False):
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    b = torch.rand(N, device=dev)
    r = torch.rand(M, N, device=dev) if residual else None
    ms = timeit(lambda: ops.gemm(a, w, bias=b, residual=r, out_dtype=out_dtype))
    print(f"gemm M={M} N={N} K={K} out={out_dtype} res={residual}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.0f} TF/s")


def attn(B, H, N):
    q = (torch.rand(B, H, N, 64, device=dev) * 2 - 1).to(torch.bfloat16)
    k = (torch.rand(B, H, N, 64, device=dev) * 2 - 1).to(torch.bfloat16)
    vt = (torch.rand(B, H, 64, N, device=dev) * 2 - 1).to(torch.bfloat16)
    ms = timeit(lambda: ops.attention(q, k, vt, N, use_exp2=True))
    print(f"attention B={B} H={H} N={N}: {ms*1e3:.1f} us  {4*B*H*N*N*64/ms/1e9:.0f} TF/s")


if __name__ == "__main__":
    gemm(4096, 4096, 4096)
    gemm(8192, 8192, 8192)
    gemm(65536, 1536, 512)
    gemm(65536, 512, 512)
    gemm(65536, 512, 512, torch.float32, True)
    gemm(65536, 512, 1408, torch.float32, True)
    gemm(65536, 2816, 512)
    gemm(65536, 8192, 512, torch.float32)
    gemm(65536, 512, 4096)
    attn(64, 8, 1024)
    attn(16, 8, 4096)
