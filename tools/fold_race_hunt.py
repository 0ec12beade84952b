"""High-volume hunt for the rare non-bit-identical result of the 256x256 GEMM variants under concurrent streams.
One GEMM variant per run, `per` launches per stream per round on `streams` streams, every output compared bit for bit with
the same call on an idle device; a mismatch is decoded (rows, columns, values, and -- for the logits variants -- which partial
sum over the K-tiles the wrong value corresponds to).

python tools/fold_race_hunt.py <variant> <rounds> [streams=3] [per=8]
variants: logits_fold logits_plain swiglu_fold swiglu_plain heads_fold heads_plain"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from paintmind_amd import ops, packing
dev = torch.device("cuda:0")
variant = sys.argv[1]
rounds = int(sys.argv[2])
nstreams = int(sys.argv[3]) if len(sys.argv) > 3 else 3
per = int(sys.argv[4]) if len(sys.argv) > 4 else 8
bf = torch.bfloat16
g = torch.Generator().manual_seed(5)
rnd = lambda *s, scale=1.0: (torch.randn(*s, generator=g) * scale).to(dev)
M, D, H = 11 * 1024, 768, 12
x = rnd(M, D) + 0.3
hi, lo = ops.split_hilo(x)
gamma, beta = 1 + 0.2 * rnd(D), 0.1 * rnd(D)
coef = ops.ln_coef(hi)
if variant.startswith("logits"):
    w = rnd(8192, D, scale=D ** -0.5); b = rnd(8192); wg, c, d = packing.ln_fold(w, gamma, beta, bf)
    fold = lambda: ops.gemm_ln(hi, wg, coef, c, d, bias=b, out_dtype=torch.float32)
    plain = lambda: ops.gemm(hi, wg, bias=b, out_dtype=torch.float32)
elif variant.startswith("swiglu"):
    w = rnd(2 * 2048, D, scale=D ** -0.5); b = rnd(2 * 2048); wg, c, d = packing.ln_fold(w, gamma, beta, bf)
    fold = lambda: ops.gemm_swiglu_ln(hi, wg, b, coef, c, d)
    plain = lambda: ops.gemm_swiglu(hi, wg, b)
else:
    w = rnd(3 * D, D, scale=D ** -0.5); wg, c, d = packing.ln_fold(w, gamma, beta, bf)
    parts = [ops.PART_Q, ops.PART_K, ops.PART_V]
    fold = lambda: torch.cat([t.flatten() for t in ops.gemm_heads_ln(hi, wg, H, 1024, parts, 0.125, coef, c, d)])
    plain = lambda: torch.cat([t.flatten() for t in ops.gemm_heads(hi, wg, H, 1024, parts, 0.125)])
MIX = variant.endswith("mix")          # fold and plain launches alternate on every stream: same box, same moment, same load
call = fold if variant.endswith("fold") else plain
ref = call()
ref_fold, ref_plain = (fold(), plain()) if MIX else (None, None)
torch.cuda.synchronize()


def decode(got, want):
    diff = (got != want)
    n = int(diff.sum())
    if got.dim() != 2:
        idx = diff.nonzero().flatten()
        print(f"   {n} elements differ; flat indices {idx[:20].tolist()}; got {got[idx[:8]].float().tolist()} want {want[idx[:8]].float().tolist()}", flush=True)
        return
    rc = diff.nonzero()
    rows, cols = rc[:, 0], rc[:, 1]
    print(f"   {n} elements differ; rows {sorted(set(rows.tolist()))[:40]} cols {sorted(set(cols.tolist()))[:40]}", flush=True)
    for r, cc in rc[:4].tolist():
        gv, wv = float(got[r, cc]), float(want[r, cc])
        line = f"   [{r},{cc}] (row%256={r % 256} col%256={cc % 256}) got {gv!r} want {wv!r}"
        if variant == "logits_fold":
            a, bb = float(coef[r, 0]), float(coef[r, 1])
            t = bb * float(c[cc]) + float(d[cc]) + float(b[cc])
            raw = (wv - t) / a
            prods = hi[r].float() * wg[cc].float()
            partial = prods.reshape(-1, 64).sum(-1).cumsum(0)
            cands = [a * float(pk) + t for pk in partial.tolist()]
            best = min(range(len(cands)), key=lambda i: abs(cands[i] - gv))
            line += f"; raw acc {raw:.6f}, rstd {a:.5f}; got==raw? {abs(gv - raw) < 1e-3 * max(1, abs(raw))}; nearest partial-K candidate: {best + 1}/{len(cands)} K-tiles ({cands[best]:.6f}); got without bias {gv - float(b[cc]):.6f}; t {t:.6f}"
        elif variant == "logits_plain":
            prods = hi[r].float() * wg[cc].float()
            partial = prods.reshape(-1, 64).sum(-1).cumsum(0)
            cands = [float(pk) + float(b[cc]) for pk in partial.tolist()]
            best = min(range(len(cands)), key=lambda i: abs(cands[i] - gv))
            line += f"; nearest partial-K candidate: {best + 1}/{len(cands)} K-tiles ({cands[best]:.6f})"
        print(line, flush=True)


streams = [torch.cuda.Stream() for _ in range(nstreams)]
bad = 0
events = []          # raw material for an offline look (tools/fold_race_decode.py)


def keep(rep, i, j, got, want):
    if got.dim() != 2 or not variant.startswith("logits"):
        return
    rc = (got != want).nonzero()
    for r, cc in rc[:6].tolist():
        r0, c0 = r // 128 * 128, cc // 64 * 64
        events.append(dict(round=rep, stream=i, launch=j, r=r, c=cc, got=float(got[r, cc]), want=float(want[r, cc]),
                           coef=coef[r0:r0 + 128].cpu(), cvec=c[c0:c0 + 64].cpu(), dvec=d[c0:c0 + 64].cpu(), bias=b[c0:c0 + 64].cpu(),
                           hi=hi[r].cpu(), w=wg[cc].cpu(), tile=want[r0:r0 + 128, c0:c0 + 64].cpu(),
                           got_tile=got[r0:r0 + 128, c0:c0 + 64].cpu()))
    torch.save(events, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"race_events_{variant}_{os.getpid()}.pt"))

t0 = time.time()
for rep in range(rounds):
    res = []
    for st in streams:
        with torch.cuda.stream(st):
            res.append([(fold if (j + rep) % 2 == 0 else plain)() for j in range(per)] if MIX else [call() for _ in range(per)])
    torch.cuda.synchronize()
    for i, outs in enumerate(res):
        for j, got in enumerate(outs):
            want = ref if not MIX else (ref_fold if (j + rep) % 2 == 0 else ref_plain)
            if not torch.equal(got, want):
                bad += 1
                kind = "" if not MIX else (" FOLD" if (j + rep) % 2 == 0 else " PLAIN")
                print(f"round {rep} stream {i} launch {j}{kind}: MISMATCH", flush=True)
                if MIX:
                    rc = (got != want).nonzero()
                    print(f"   {rc.shape[0]} elements differ; rows%256 {sorted(set((rc[:, 0] % 256).tolist()))} cols%256 {sorted(set((rc[:, 1] % 256).tolist()))}; "
                          f"max abs diff {float((got - want).abs().max()):.5f}", flush=True)
                else:
                    decode(got, want)
                    keep(rep, i, j, got, want)
print(f"variant={variant} streams={nstreams}: {bad} bad of {rounds * nstreams * per} launches, {time.time() - t0:.0f} s "
      f"(PERSIST256={os.environ.get('PMHIP_PERSIST256', 'default')})", flush=True)
