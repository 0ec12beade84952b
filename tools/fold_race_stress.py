"""operator-level stress: the hi/lo producers, the coefficient kernel and the LayerNorm-folded consumers on three streams at once,
every result compared bit for bit with the same call on an idle device.  python tools/fold_race_stress.py [reps] [which]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paintmind_amd import ops, packing
dev = torch.device("cuda:0")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
which = sys.argv[2] if len(sys.argv) > 2 else "all"
bf = torch.bfloat16
g = torch.Generator().manual_seed(5)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
M, D, H = 11 * 1024, 768, 12
x = rnd(M, D) + 0.3
hi, lo = ops.split_hilo(x)
gamma, beta = 1 + 0.2 * rnd(D), 0.1 * rnd(D)
wq = rnd(3 * D, D, scale=D ** -0.5); wgq, cq, dq = packing.ln_fold(wq, gamma, beta, bf)
w12 = rnd(2 * 2048, D, scale=D ** -0.5); b12 = rnd(2 * 2048); w12g, c12, d12 = packing.ln_fold(w12, gamma, beta, bf)
wl = rnd(8192, D, scale=D ** -0.5); bl = rnd(8192); wlg, cl, dl = packing.ln_fold(wl, gamma, beta, bf)
a = rnd(M, D, scale=0.5).to(bf); wo = rnd(D, D, scale=D ** -0.5).to(bf); bo = rnd(D)
hid = rnd(M, 2048, scale=0.5).to(bf); w3 = rnd(D, 2048, scale=2048 ** -0.5).to(bf)
q = rnd(11, H, 1024, 64, scale=0.6).to(bf); k = rnd(11, H, 1024, 64).to(bf); vt = rnd(11, H, 64, 1024).to(bf)

def work(i):
    out = []
    if which in ("all", "coef"):
        out.append(ops.ln_coef(hi))
    coef = ops.ln_coef(hi)
    if which in ("all", "heads"):
        out += list(ops.gemm_heads_ln(hi, wgq, H, 1024, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.125, coef, cq, dq))
    if which in ("all", "qonly"):
        out += list(ops.gemm_heads_ln(hi, wgq[:D].contiguous(), H, 1024, [ops.PART_Q], 0.125, coef, cq[:D].contiguous(), dq[:D].contiguous()))
    if which in ("all", "swiglu"):
        out.append(ops.gemm_swiglu_ln(hi, w12g, b12, coef, c12, d12))
    if which in ("all", "logits"):
        out.append(ops.gemm_ln(hi, wlg, coef, cl, dl, bias=bl, out_dtype=torch.float32))
    if which in ("all", "prod"):
        out += list(ops.gemm_hilo(a, wo, hi, lo, bias=bo))
        out += list(ops.gemm_hilo(hid, w3, hi, lo, bias=bo))
    if which in ("all", "attn"):
        out.append(ops.attention(q, k, vt, 1024, use_exp2=True))
    return out

ref = work(0)
torch.cuda.synchronize()
streams = [torch.cuda.Stream() for _ in range(3)]
bad = {}
for rep in range(reps):
    res = []
    for st in streams:
        with torch.cuda.stream(st):
            res.append(work(0))
    torch.cuda.synchronize()
    for i in range(3):
        for j, (got, want) in enumerate(zip(res[i], ref)):
            if not torch.equal(got, want):
                d = (got.float() - want.float()).abs()
                bad[j] = bad.get(j, 0) + 1
                rows = d.reshape(d.shape[0] if d.dim() == 2 else d.shape[0] * d.shape[1], -1).amax(-1).nonzero().flatten()
                print(f"rep {rep} stream {i} output {j} shape {tuple(got.shape)}: {int((d > 0).sum())} elements differ, max {float(d.max()):.5f}, rows {rows[:6].tolist()}..{rows[-2:].tolist()} ({rows.numel()})", flush=True)
print(f"which={which}: mismatches per output index {bad} over {reps} x 3 runs", flush=True)
