"""Measurement behind the bf16 ViT gates of tests/test_gpu_model.py and bench.py (RECON_GATE): bf16 mode against fp32-verify on the
vit-s towers, several seeds at B = 4 and the bench input at B = 64.  python tools/bf16_vit_stats.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch

import paintmind_amd as pm
from test_gpu_model import _bf16_vit_stats

dev = torch.device("cuda:0")
torch.manual_seed(0)
m = pm.create_model(arch="vqgan", version="vit-s-vqgan", pretrained=False).to(dev).eval()
for B, seeds in ((4, range(100, 108)), (64, (0, 1, 2))):
    for seed in seeds:
        x = (torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(seed)) * 2 - 1).to(dev)
        agree, dmean, dmax, down = _bf16_vit_stats(m, x)
        print(f"B={B} seed={seed}: token agreement {agree:.5f} rec |dev| mean {dmean:.5f} max {dmax:.4f} own-token mean {down:.5f}", flush=True)
