"""per-K-tile cost of the small-batch GEMM forms: the residual producer at M = 1024 (one image), N = 512, K swept; hipGraph-free, one
kernel per launch, timed by hipEvents over a dependent chain of launches.   python tools/small_gemm_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from paintmind_amd import _lib, ops

dev = torch.device("cuda:0")
lib = _lib.load()
bf = torch.bfloat16
M, N = int(os.environ.get("PROBE_M", "1024")), 512
res = {}
for K in (256, 512, 1024, 1408, 2048, 4096):
    a = (torch.randn(M, K, device=dev) * 0.5).to(bf)
    w = (torch.randn(N, K, device=dev) * K ** -0.5).to(bf)
    b = torch.randn(N, device=dev)
    hi, lo = ops.split_hilo(torch.randn(M, N, device=dev))
    parts = torch.empty(M, N // 64, 2, device=dev)
    s = ops.stream_ptr(dev)
    run = lambda: lib.pmhip_gemm_hilo_stats(a.data_ptr(), K, w.data_ptr(), K, b.data_ptr(), hi.data_ptr(), lo.data_ptr(), N, 0, hi.data_ptr(),
                                            lo.data_ptr(), N, M, N, K, parts.data_ptr(), s)
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        run()
    e1.record()
    torch.cuda.synchronize()
    res[K] = e0.elapsed_time(e1) / 200 * 1e3
    print(f"M={M} N={N} K={K:5d} ({K // 64:3d} K-tiles): {res[K]:7.2f} us per launch (back to back, in place)", flush=True)
ks = sorted(res)
print(f"slope {ks[-1]} vs {ks[1]}: {(res[ks[-1]] - res[ks[1]]) / ((ks[-1] - ks[1]) / 64):.3f} us per K-tile; intercept ~ {res[ks[1]] - (res[ks[-1]] - res[ks[1]]) / ((ks[-1] - ks[1]) / 64) * (ks[1] / 64):.2f} us")
