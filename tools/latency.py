"""B=1..8 generate latency, eager vs hipGraph replay"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import paintmind_amd as pm
from paintmind_amd.config import ver2cfg
from paintmind_amd.generate import Pipeline
dev = torch.device("cuda:0")
torch.manual_seed(0)
pipe = Pipeline(pm.Config(ver2cfg["bench-uncond-12L-d512"]), stage1_pretrained=False).to(dev).eval()
pipe.set_compute_dtype(torch.bfloat16)
for B in (1, 2, 8):
    for graph in (False, True):
        for i in range(3):
            pipe.generate_ids(None, B, 8, 1.0, 5, [True] * 8, seed=i, use_graph=graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 5
        for i in range(n):
            pipe.generate_ids(None, B, 8, 1.0, 5, [True] * 8, seed=10 + i, use_graph=graph)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"B={B} graph={graph}: {dt*1e3:.1f} ms per 8-step generate -> {B/dt:.1f} img/s")
