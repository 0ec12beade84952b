"""bf16 logits of the 12L/d512 transformer against fp32-verify over 16 images: run with PMHIP_F32_STREAM=1, PMHIP_LN_UNFOLD=1
and with neither (fold) to compare the three residual-stream / LayerNorm arrangements on identical inputs"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import paintmind_amd as pm
from paintmind_amd.config import ver2cfg
from paintmind_amd.generate import Pipeline
dev = torch.device("cuda:0")
torch.manual_seed(0)
pipe = Pipeline(pm.Config(ver2cfg["bench-uncond-12L-d512"]), stage1_pretrained=False).to(dev).eval()
B = 16
ids = torch.randint(0, 8192, (B, 1024), generator=torch.Generator().manual_seed(1)).to(dev)
ids[:, ::3] = 8192
tok = pipe.ids2tokens(ids)
l32 = torch.cat([pipe.tokens2logits(tok[i:i + 4], None) for i in range(0, B, 4)])
pipe.set_compute_dtype(torch.bfloat16)
l16 = pipe.tokens2logits(tok, None)
err = (l16 - l32).abs()
top2 = torch.topk(l32, 2, dim=-1).values
gap = top2[..., 0] - top2[..., 1]
flips = l16.argmax(-1) != l32.argmax(-1)
print(f"{os.environ.get('PMHIP_F32_STREAM','0')}/{os.environ.get('PMHIP_LN_UNFOLD','0')}: max err {float(err.max()):.5f} mean err {float(err.mean()):.6f} "
      f"agreement {1 - float(flips.float().mean()):.4f} ({int(flips.sum())} flips of {flips.numel()}); flips with gap > 0.01: {int((gap[flips] > 0.01).sum())}")
