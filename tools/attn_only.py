import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paintmind_amd import ops
dev = torch.device("cuda:0")
B, H, N = 64, 8, 1024
q = (torch.rand(B, H, N, 64, device=dev) * 2 - 1).to(torch.bfloat16)
k = (torch.rand(B, H, N, 64, device=dev) * 2 - 1).to(torch.bfloat16)
vt = (torch.rand(B, H, 64, N, device=dev) * 2 - 1).to(torch.bfloat16)
for _ in range(5):
    ops.attention(q, k, vt, N, use_exp2=True)
torch.cuda.synchronize()
