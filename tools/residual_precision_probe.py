"""What would a narrower residual stream cost in accuracy?  Operator-level forward of the 12L/d512 transformer in bf16 mode
with the residual GEMMs' fp32 output rounded (a) not at all, (b) to bf16, (c) to a bf16 hi/lo pair; logits against fp32-verify."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import paintmind_amd as pm
from paintmind_amd import ops
from paintmind_amd.config import ver2cfg
from paintmind_amd.generate import Pipeline
dev = torch.device("cuda:0")
torch.manual_seed(0)
pipe = Pipeline(pm.Config(ver2cfg["bench-uncond-12L-d512"]), stage1_pretrained=False).to(dev).eval()
B = 4
ids = torch.randint(0, 8192, (B, 1024), generator=torch.Generator().manual_seed(1)).to(dev)
ids[:, ::3] = 8192
tok = pipe.ids2tokens(ids)
l32 = pipe.transformer(tok, None)
orig = ops.gemm
mode = [None]
def patched(a, w, bias=None, residual=None, res_rows=0, out_dtype=None, **kw):
    out = orig(a, w, bias=bias, residual=residual, res_rows=res_rows, out_dtype=out_dtype, **kw)
    if residual is not None and mode[0] == "bf16":
        out = out.to(torch.bfloat16).float()
    elif residual is not None and mode[0] == "hilo":
        hi = out.to(torch.bfloat16).float()
        out = hi + (out - hi).to(torch.bfloat16).float()
    return out
ops.gemm = patched
import paintmind_amd.stage2.transformer as TR, paintmind_amd.modules.attention as AT, paintmind_amd.modules.mlp as ML
pipe.set_compute_dtype(torch.bfloat16)
for m in (None, "bf16", "hilo"):
    mode[0] = m
    l16 = pipe.transformer(tok, None)
    err = (l16 - l32).abs()
    agree = float((l16.argmax(-1) == l32.argmax(-1)).float().mean())
    cos = torch.nn.functional.cosine_similarity(l16, l32, dim=-1)
    print(f"residual {str(m):5s}: logits max err {float(err.max()):.5f} mean err {float(err.mean()):.6f} cos min {float(cos.min()):.6f} top-1 agreement {agree:.4f}")
