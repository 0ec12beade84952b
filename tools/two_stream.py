"""experiment: one 64-image generate vs two concurrent 32-image generates on separate streams (graph replay)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import paintmind_amd as pm
from paintmind_amd.config import ver2cfg
from paintmind_amd.generate import Pipeline

dev = torch.device("cuda:0")
def mk():
    torch.manual_seed(0)
    p = Pipeline(pm.Config(ver2cfg["bench-uncond-12L-d512"]), stage1_pretrained=False).to(dev).eval()
    p.set_compute_dtype(torch.bfloat16)
    return p
T = 8
def runner(pipe, B, stream):
    eng, vq = pipe.engine(), pipe.vqgan.engine()
    temps, nmask = pipe._schedule(T, 1.0)
    flags = [True] * T
    def go(i):
        with torch.cuda.stream(stream):
            ids = torch.full((B, 1024), 8192, dtype=torch.long, device=dev)
            eng.generate(vq, ids, None, temps, nmask, flags, topk=5, seed=i, use_graph=True)
    return go
def bench(fns, iters=4):
    for i in range(3):
        for f in fns: f(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(iters):
        for f in fns: f(10 + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters
pipes = [mk() for _ in range(8)]
for k, nb in ((1, 64), (4, 16), (8, 8), (4, 16), (3, 21), (2, 32), (1, 64)):
    t = bench([runner(pipes[i], nb, torch.cuda.Stream()) for i in range(k)])
    print(f"{k} streams x B={nb}: {t*1e3:.1f} ms -> {k*nb/t:.1f} img/s")
