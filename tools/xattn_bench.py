import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.getcwd())
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
B, H, N = 32, 12, 1024
q = (torch.randn(B, H, N, 64, device=dev) * 0.5).to(torch.bfloat16)
for nkv, pad in ((1024, 1024), (77, 128), (64, 64), (128, 128), (192, 192)):
    k = torch.randn(B, H, pad, 64, device=dev).to(torch.bfloat16); vt = torch.randn(B, H, 64, pad, device=dev).to(torch.bfloat16)
    ms = timeit(lambda: ops.attention(q, k, vt, nkv, use_exp2=True))
    print(f"B={B} H={H} Nq={N} Nkv={nkv}: {ms*1e3:.1f} us  {4*B*H*N*nkv*64/ms/1e9:.0f} TF/s")
