"""three drop-in generate() calls for a rocprofv3 kernel + memory-copy trace (how the D2H copies sit under the lanes)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
dev = torch.device("cuda:0")
pipe = bench.build(bench.DEFAULT_WORKLOAD, dev, torch.bfloat16)
from paintmind_amd.modules.encoder import NullTextEmbedder
pipe.text_model = NullTextEmbedder()
text = ["p"] * 64
kw = dict(streams=int(os.environ.get("LANES", "2")))
for i in range(5):
    t0 = time.perf_counter(); pipe.generate(text, timesteps=8, topk=5, save_interval=1, seed=i, **kw); print(i, (time.perf_counter() - t0) * 1e3)
