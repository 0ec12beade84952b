"""What the epilogues of the K = 512 GEMMs cost (round 6): the head-split QKV GEMM and the SwiGLU GEMM at the bench's launch shape,
timed with the library as built.  Run under tools/ab_flags.sh with "" / -DPM_ABL_EPI_SKIP (gemm_common.h: no epilogue at all):
    gpurun -- 'bash tools/ab_flags.sh "python tools/epilogue_cost.py" "" -DPM_ABL_EPI_SKIP'"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paintmind_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


M, D, H, N = 32768, 512, 8, 1024
g = torch.Generator().manual_seed(0)
x = (torch.randn(M, D, generator=g)).to(dev).to(torch.bfloat16)
wqkv = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).to(torch.bfloat16)
w12 = (torch.randn(2 * 1408, D, generator=g) * D ** -0.5).to(dev).to(torch.bfloat16)
b12 = torch.randn(2 * 1408, generator=g).to(dev)
ms = timeit(lambda: ops.gemm_heads(x, wqkv, H, N, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.18))
print(f"QKV head-split  M={M} N={3 * D} K={D}: {ms * 1e3:7.1f} us  {2 * M * 3 * D * D / ms / 1e9:6.0f} TFLOP/s")
ms = timeit(lambda: ops.gemm_swiglu(x, w12, b12))
print(f"SwiGLU          M={M} N={2 * 1408} K={D}: {ms * 1e3:7.1f} us  {2 * M * 2 * 1408 * D / ms / 1e9:6.0f} TFLOP/s")
