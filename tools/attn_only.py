"""attention kernel alone at the bench shape (B=64, H=8, N=1024, dh=64, bf16, exp2): TFLOP/s by hipEvents, plus a
float64 check of one head.  PMHIP_ATTN_OLD=1 selects the round-2 kernel (attention.hip) instead of attention_bf16.hip."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paintmind_amd import ops
dev = torch.device("cuda:0")
B, H, N = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (64, 8, 1024)))
g = torch.Generator().manual_seed(0)
q = ((torch.rand(B, H, N, 64, generator=g) * 2 - 1) * 0.5).to(dev).to(torch.bfloat16)
k = (torch.rand(B, H, N, 64, generator=g) * 2 - 1).to(dev).to(torch.bfloat16)
vt = (torch.rand(B, H, 64, N, generator=g) * 2 - 1).to(dev).to(torch.bfloat16)
for _ in range(5):
    out = ops.attention(q, k, vt, N, use_exp2=True)
torch.cuda.synchronize()
# correctness of (b=1, h=3) against float64
b, h = min(1, B - 1), min(3, H - 1)
s = (q[b, h].double() @ k[b, h].double().T) * 0.6931471805599453
p = torch.softmax(s, -1)
ref = p @ vt[b, h].double().T
got = out.reshape(B, N, H, 64)[b, :, h].double()
print("max abs err vs float64:", float((got - ref).abs().max()), "ref max", float(ref.abs().max()))
reps = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.attention(q, k, vt, N, use_exp2=True)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"B={B} H={H} N={N}: {ms * 1e3:.1f} us  {4 * N * N * 64 * B * H / ms / 1e9:.1f} TFLOP/s  (ATTN_OLD={os.environ.get('PMHIP_ATTN_OLD', '0')})")
