"""Bandwidth of the masked cross-entropy / random-masking kernels (loss.hip) at the bench shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paintmind_amd import ops

dev = torch.device("cuda:0")
M, V = 65536, 8192
logits = torch.randn(M, V, device=dev)
labels = torch.randint(0, V, (M,), device=dev)
mask = (torch.rand(M, device=dev) < 0.5).float()
z = torch.randn(64, 1024, 32, device=dev)
noise = torch.rand(64, 1024, device=dev)
tok = torch.randn(32, device=dev)
for name, fn, nbytes in (("masked_ce", lambda: ops.masked_ce(logits, labels, mask, 0.1), M * V * 4),
                         ("random_mask", lambda: ops.random_mask(z, noise, tok, 256), 2 * z.numel() * 4)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name}: {ms*1e3:.1f} us  {nbytes/ms/1e6:.0f} GB/s")
