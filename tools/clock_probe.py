"""What clock does GRBM_GUI_ACTIVE tick at, and does it drop under matrix load?  Run under
   rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d <dir> -o p -- python3 tools/clock_probe.py
then divide each kernel's GRBM_GUI_ACTIVE / 8 by its duration (tools/clock_probe_report.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paintmind_amd import ops
dev = torch.device("cuda:0")
x = torch.zeros(256 * 1024 * 1024, device=dev)            # 1 GiB stream
a = (torch.rand(8192, 8192, device=dev) * 2 - 1).to(torch.bfloat16)
w = (torch.rand(8192, 8192, device=dev) * 2 - 1).to(torch.bfloat16)
z = torch.zeros(8192, 8192, device=dev, dtype=torch.bfloat16)
q = ((torch.rand(64, 8, 1024, 64, device=dev) * 2 - 1) * 0.5).to(torch.bfloat16)
k = (torch.rand(64, 8, 1024, 64, device=dev) * 2 - 1).to(torch.bfloat16)
vt = (torch.rand(64, 8, 64, 1024, device=dev) * 2 - 1).to(torch.bfloat16)
for rep in range(3):
    for _ in range(4):
        x.add_(1.0)                                        # HBM-bound
    for _ in range(6):
        ops.gemm(a, w, out_dtype=torch.bfloat16)           # MFMA-bound, random data
    for _ in range(6):
        ops.gemm(z, z, out_dtype=torch.bfloat16)           # MFMA-bound, zero data
    for _ in range(12):
        ops.attention(q, k, vt, 1024, use_exp2=True)        # attention at the bench shape, random data
torch.cuda.synchronize()
