"""does handing the logits to the sampler through the caches (row chunks of a few images, GEMM and sampler
alternating) beat one full-size logits GEMM + one full-size sample_rows?  python tools/logits_chunk_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from paintmind_amd import ops

dev = torch.device("cuda:0")
B, N, D, V = 64, 1024, 512, 8192
M = B * N
bf = torch.bfloat16
a = (torch.randn(M, D, device=dev) * 0.7).to(bf)
w = (torch.randn(V, D, device=dev) * D ** -0.5).to(bf)
bias = torch.randn(V, device=dev) * 0.1
ids = torch.full((M,), V, dtype=torch.long, device=dev)


def run(chunk_images):
    rows = chunk_images * N
    outs = []
    for m0 in range(0, M, rows):
        lg = ops.gemm(a[m0:m0 + rows], w, bias=bias, out_dtype=torch.float32)
        outs.append(ops.sample_rows(lg, ids[m0:m0 + rows], V, 5, 0.7, seed=3, step=1, row_base=m0))
    return outs


def timeit(fn, iters=10):
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


ref = run(64)
for ci in (64, 32, 16, 8, 4, 2):
    got = run(ci)
    pred = torch.cat([g[0] for g in got])
    same = bool(torch.equal(pred, ref[0][0]))
    for _ in range(2):
        ms = timeit(lambda: run(ci))
        print(f"chunk of {ci:2d} images ({ci * N * V * 4 / 2**20:6.0f} MiB of logits per chunk): {ms*1e3:7.1f} us per step for GEMM + sampling, identical ids: {same}", flush=True)
