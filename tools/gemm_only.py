"""one GEMM shape a few times (for rocprofv3 PMC passes): python tools/gemm_only.py M N K"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paintmind_amd import ops
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8192, 8192, 8192)
dev = torch.device("cuda:0")
a = (torch.rand(M, K, device=dev) * 2 - 1).to(torch.bfloat16)
w = (torch.rand(N, K, device=dev) * 2 - 1).to(torch.bfloat16)
b = torch.rand(N, device=dev)
for _ in range(4):
    ops.gemm(a, w, bias=b, out_dtype=torch.bfloat16)
torch.cuda.synchronize()
