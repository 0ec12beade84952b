"""QKV head-split and SwiGLU GEMM time per image against the batch size (round 6): the persistent 256 x 256 kernel walks
tiles / 256 rounds, so a batch whose tile count is not a multiple of 256 pays for a partly empty last round -- B = 33 costs 18 %
more per image than B = 32 or 64 (QKV), B = 31 14 % more (SwiGLU).  bench.py's two lanes of 33 + 31 images each pay that on their
own, and absorb it by running beside each other (2 lanes == 1 lane of 64 within the spread: tools/ab_env.sh PM_BENCH_STREAMS).
    gpurun -- 'python tools/tile_quantisation_probe.py'"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paintmind_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
D, H, N = 512, 8, 1024
g = torch.Generator().manual_seed(0)
wqkv = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).to(torch.bfloat16)
w12 = (torch.randn(2 * 1408, D, generator=g) * D ** -0.5).to(dev).to(torch.bfloat16)
b12 = torch.randn(2 * 1408, generator=g).to(dev)
for B in (64, 48, 40, 36, 33, 32, 31, 28, 24, 16):
    M = B * 1024
    x = torch.randn(M, D, generator=g).to(dev).to(torch.bfloat16)
    a = timeit(lambda: ops.gemm_heads(x, wqkv, H, N, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.18))
    b = timeit(lambda: ops.gemm_swiglu(x, w12, b12))
    tq, ts = B * 4 * 6, B * 4 * 11
    print(f"B={B:3d}  QKV {a*1e3:7.1f} us  {a*1e3/B:6.3f} us/img  tiles {tq:5d} = {tq/256:5.2f} rounds | SwiGLU {b*1e3:7.1f} us {b*1e3/B:6.3f} us/img tiles {ts:5d} = {ts/256:5.2f} rounds")
