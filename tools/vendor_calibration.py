"""Calibration, not product: what the vendor libraries of this image (hipBLASLt behind torch.mm, the SDPA flash kernel) reach
on the bench shapes, next to this library's kernels ON THE SAME BOX.  Tells how much head-room a shape has before more tuning
is attempted.  python tools/vendor_calibration.py"""
import os
import sys

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


def gemm(M, N, K, out_dtype=torch.bfloat16):
    a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
    w = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    ms_v = timeit(lambda: torch.mm(a, w.t(), out=out))
    ms_o = timeit(lambda: ops.gemm(a, w, out_dtype=out_dtype))
    fl = 2 * M * N * K
    print(f"gemm M={M} N={N} K={K}: vendor(bf16 out) {ms_v*1e3:.1f} us {fl/ms_v/1e9:.0f} TF/s | ours({out_dtype}) {ms_o*1e3:.1f} us {fl/ms_o/1e9:.0f} TF/s", flush=True)


def attn(B, H, N):
    q = (torch.rand(B, H, N, 64, device=dev) * 2 - 1).to(torch.bfloat16)
    k = (torch.rand(B, H, N, 64, device=dev) * 2 - 1).to(torch.bfloat16)
    v = (torch.rand(B, H, N, 64, device=dev) * 2 - 1).to(torch.bfloat16)
    vt = v.transpose(2, 3).contiguous()
    fl = 4 * B * H * N * N * 64
    ms_o = timeit(lambda: ops.attention(q, k, vt, N, use_exp2=True))
    line = f"attention B={B} H={H} N={N}: ours {ms_o*1e3:.1f} us {fl/ms_o/1e9:.0f} TF/s"
    from torch.nn.attention import SDPBackend, sdpa_kernel
    for name, be in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION)):
        try:
            with sdpa_kernel(be):
                ms_v = timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v))
            line += f" | sdpa-{name} {ms_v*1e3:.1f} us {fl/ms_v/1e9:.0f} TF/s"
        except Exception as e:
            line += f" | sdpa-{name} unavailable ({type(e).__name__})"
    print(line, flush=True)


if __name__ == "__main__":
    for _ in range(2):
        gemm(8192, 8192, 8192)
        gemm(65536, 1536, 512)
        gemm(65536, 2816, 512)
        gemm(65536, 512, 512)
        gemm(65536, 512, 1408)
        gemm(65536, 8192, 512, torch.float32)
        attn(64, 8, 1024)
        attn(16, 8, 4096)
