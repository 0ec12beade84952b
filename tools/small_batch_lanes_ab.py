"""Mid-size batches (B = 4..64, AB_BATCHES): one lane (its decode deferred beside the next step, engine.hip) against two lanes
(which never defer their decode since round 5) against two lanes that do (the behaviour before the fix, by monkeypatch).
ms per 8-step generate of the headline model, hipGraph replay; alternating, best of 3 blocks of 5 calls."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import paintmind_amd as pm
from paintmind_amd.config import ver2cfg
from paintmind_amd.generate import Pipeline
from paintmind_amd import engine as E
cfg = sys.argv[1] if len(sys.argv) > 1 else "bench-uncond-12L-d512"
dev = torch.device("cuda:0")
torch.manual_seed(0)
text_model = None
if "text_model" not in ver2cfg[cfg]:                       # a reference preset: its T5 tower is a download, stand in for it
    from paintmind_amd.modules.encoder import SyntheticTextEmbedder
    from bench import context_dim_of
    text_model = SyntheticTextEmbedder(context_dim_of(cfg))
pipe = Pipeline(pm.Config(ver2cfg[cfg]), stage1_pretrained=False, text_model=text_model).to(dev).eval()
pipe.set_compute_dtype(torch.bfloat16)
from bench import context_dim_of
ctx_dim = None if cfg == "bench-uncond-12L-d512" else context_dim_of(cfg)
cls = E.S2Engine
orig = cls.generate


def deferring(self, *a, **kw):
    kw["concurrent_lanes"] = False
    return orig(self, *a, **kw)


def timed(B, streams, ctx):
    for i in range(3):
        pipe.generate_ids(ctx, B, 8, 1.0, 5, [True] * 8, seed=i, use_graph=True, streams=streams)
    best = 1e9
    for blk in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(5):
            pipe.generate_ids(ctx, B, 8, 1.0, 5, [True] * 8, seed=10 + i, use_graph=True, streams=streams)
            torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 5)
    return best * 1e3


for B in [int(x) for x in os.environ.get("AB_BATCHES", "4,8,12,16,24,32").split(",")]:
    ctx = None if ctx_dim is None else torch.randn(B, 77, ctx_dim, device=dev)
    one = timed(B, 1, ctx)
    two = timed(B, 2, ctx)
    cls.generate = deferring
    old = timed(B, 2, ctx)
    cls.generate = orig
    print(f"{cfg} B={B:3d}: one lane {one:7.2f} ms | two lanes {two:7.2f} ms | two lanes, decode deferred (pre-fix) {old:7.2f} ms", flush=True)
