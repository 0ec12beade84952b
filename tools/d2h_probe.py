"""host-side costs of the drop-in generate(): pinned allocation, D2H bandwidth, and the call with / without host copies"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
x = torch.randn(8, 64, 3, 256, 256, device=dev)
for i in range(3):
    t0 = time.perf_counter(); h = torch.empty(x.shape, dtype=torch.float32, pin_memory=True); t1 = time.perf_counter()
    print(f"pinned alloc {x.numel()*4/1e6:.0f} MB: {(t1-t0)*1e3:.2f} ms")
    torch.cuda.synchronize(); t0 = time.perf_counter(); h.copy_(x, non_blocking=True); torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"  D2H: {(t1-t0)*1e3:.2f} ms = {x.numel()*4/(t1-t0)/1e9:.1f} GB/s")
    del h
import bench
pipe = bench.build(bench.DEFAULT_WORKLOAD, dev, torch.bfloat16)
from paintmind_amd.modules.encoder import NullTextEmbedder
pipe.text_model = NullTextEmbedder()
text = ["p"] * 64
for kw in (dict(keep_on_device=True), dict(), dict(streams=(32, 32)), dict(streams=(34, 30)), dict(streams=1)):
    for si in (1, 2):
        for i in range(3): pipe.generate(text, timesteps=8, topk=5, save_interval=si, seed=i, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(4): out = pipe.generate(text, timesteps=8, topk=5, save_interval=si, seed=10 + i, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4
        print(kw, "save_interval", si, f"{dt*1e3:.1f} ms/call  {64/dt:.1f} img/s")
