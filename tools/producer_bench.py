"""the residual producers alone at the bench shape (in place, with row statistics, as the engine runs them):
out-projection (gemm2b, K = 512) and FFN w3 (gemm256, K = 1408).  python tools/producer_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C

import torch

from paintmind_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()


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


def producer(M, N, K, rounds=2):
    bf = torch.bfloat16
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(bf)
    w = ((torch.rand(N, K, device=dev) * 2 - 1) * K ** -0.5).to(bf)
    b = torch.rand(N, device=dev)
    hi, lo = ops.split_hilo(torch.randn(M, N, device=dev))
    parts = torch.empty(M, N // 64, 2, device=dev)
    s = ops.stream_ptr(dev)

    def run():
        lib.pmhip_gemm_hilo_stats(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), hi.data_ptr(), lo.data_ptr(), N, 0, hi.data_ptr(), lo.data_ptr(),
                                  N, M, N, K, parts.data_ptr(), s)
    for _ in range(rounds):
        ms = timeit(run)
        mb = M * (K * 2 + N * 8) / 1e6
        print(f"producer M={M} N={N} K={K}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.0f} TF/s  {mb/ms/1e3:.2f} TB/s algorithmic", flush=True)


if __name__ == "__main__":
    producer(65536, 512, 512)
    producer(65536, 512, 1408)
    producer(32768, 512, 512, 1)
    producer(16384, 512, 512, 1)
