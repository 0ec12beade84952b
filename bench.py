#!/usr/bin/env python3
"""Headline benchmark: 256x256 images/sec/GPU, 8-step MaskGIT decode with vit-s-vqgan tokens.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path over one batch of synthetic input that is already resident
in HBM: for the default workload (BASELINE.json configs[2]) that is ONE Pipeline.generate() of B=64
images -- 8 MaskGIT steps of the 12-layer d=512 transformer (context=None, so attn2 is a second
self-attention) plus a ViT decode of the sampled tokens after EVERY step, which is the work the
reference does per generate() (generate.py:165; nothing is skipped).  Images stay on the device.
Multi-GPU: weak scaling, every rank decodes its own 64 images (RNG keyed by global image index), no
collective in the data path, one RCCL gather of the finished images to rank 0 inside the timed region.

One JSON line is printed by rank 0; see DESIGN.md for how `roofline` and `cpu_baseline` are derived.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import paintmind_amd as pm  # noqa: E402
from paintmind_amd import ops  # noqa: E402
from paintmind_amd.config import ver2cfg  # noqa: E402
from paintmind_amd.generate import Pipeline  # noqa: E402
from paintmind_amd.ops import swiglu_hidden  # noqa: E402

USE_GRAPH = os.environ.get("PM_BENCH_NO_GRAPH", "0") != "1"   # decode loop = one replayed hipGraph (captured during warm-up)
STREAMS = int(os.environ.get("PM_BENCH_STREAMS", "3"))   # concurrent micro-batches per GPU (1 = one stream)
LANE_SPLIT = os.environ.get("PM_BENCH_LANE_SPLIT")       # development: explicit micro-batch sizes, e.g. "32,16,16"
PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16, MI355X_MICROARCH.md chip table
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0

WORKLOADS = {
    # name: (pipeline config, batch per GPU, timesteps, context length or None)
    "maskgit-uncond-12L-d512-T8": ("bench-uncond-12L-d512", 64, 8, None),      # BASELINE configs[2]
    "maskgit-text-24L-d768-T8": ("bench-text-24L-d768", 32, 8, 77),            # north_star target model, 8 steps
    "maskgit-text-24L-d768-T12": ("bench-text-24L-d768", 32, 12, 77),          # BASELINE configs[3] per-GPU share
    "vit-s-recon": (None, 64, 0, None),                                         # BASELINE configs[1]
}


# ---------------------------------------------------------------------------------------------
# algorithmic work (2*M*N*K per GEMM, 4*N*Nkv*inner per attention core)
# ---------------------------------------------------------------------------------------------
def layer_flops(D, heads, mlp_dim, N, stage2, ctx_len):
    inner, hf = heads * 64, swiglu_hidden(mlp_dim)
    proj_self = 4 * 2 * N * D * inner
    core_self = 4 * N * N * inner
    gemm, attn = proj_self, core_self
    if stage2:
        if ctx_len is None:
            gemm += proj_self
            attn += core_self
        else:
            gemm += 2 * 2 * N * D * inner + 2 * 2 * ctx_len * D * inner
            attn += 4 * N * ctx_len * inner
    gemm += 6 * N * D * hf
    return gemm, attn


def vit_flops(tower, embed_dim, patch_k, encode):
    N = (tower["image_size"] // tower["patch_size"]) ** 2
    g, a = layer_flops(tower["dim"], tower["num_head"], tower["mlp_dim"], N, False, None)
    gemm, attn = g * tower["depth"], a * tower["depth"]
    gemm += 2 * N * patch_k * tower["dim"] + 2 * N * tower["dim"] * embed_dim
    if encode:
        gemm += 2 * N * embed_dim * 8192          # the VQ distance product (quantize.py:26)
    return gemm, attn


def s2_step_flops(cfg, N, embed_dim, n_embed, ctx_len):
    g, a = layer_flops(cfg["dim"], cfg["num_head"], cfg["mlp_dim"], N, True, ctx_len)
    gemm, attn = g * cfg["depth"], a * cfg["depth"]
    gemm += 2 * N * embed_dim * cfg["dim"] + 2 * N * cfg["dim"] * n_embed
    return gemm, attn


# ---------------------------------------------------------------------------------------------
def build(workload, device, dtype):
    cfg_name, B, T, L = WORKLOADS[workload]
    torch.manual_seed(0)
    if cfg_name is None:
        model = pm.create_model(arch="vqgan", version="vit-s-vqgan", pretrained=False).to(device).eval()
        model.set_compute_dtype(dtype)
        return model, None
    pipe = Pipeline(pm.Config(ver2cfg[cfg_name]), stage1_pretrained=False).to(device).eval()
    pipe.set_compute_dtype(dtype)
    return pipe, ver2cfg[cfg_name]


def make_step(workload, model, device, rank, decode_every_step=True):
    cfg_name, B, T, L = WORKLOADS[workload]
    if cfg_name is None:
        x = (torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(rank)) * 2 - 1).to(device)

        def step(i):
            z, _, _ = model.encode(x)
            return model.decode(z)
        return step
    pipe = model
    ctx = None
    if L is not None:
        g = torch.Generator().manual_seed(1234 + rank)
        ctx = torch.randn(B, L, ver2cfg[cfg_name]["context_dim"], generator=g).to(device)
    flags = [True] * T if decode_every_step else [t == T - 1 for t in range(T)]

    def step(i, join=True, streams=None):
        STREAMS = globals()["STREAMS"] if streams is None else streams
        lanes = STREAMS
        if LANE_SPLIT and STREAMS > 1:
            lanes = tuple(int(x) for x in LANE_SPLIT.split(","))
        # join=False (single-GPU timed loop): the micro-batch lanes are not joined between steps, so consecutive
        # steps pipeline across lanes; the caller joins once before the closing synchronize
        if STREAMS > 1 and not join:
            return pipe.generate_ids(ctx, B, T, 1.0, 5, flags, seed=1000 + i, image_base=rank * B, use_graph=USE_GRAPH,
                                     streams=lanes, join=False, wait_current=False)
        ids, imgs = pipe.generate_ids(ctx, B, T, 1.0, 5, flags, seed=1000 + i, image_base=rank * B, use_graph=USE_GRAPH,
                                      streams=lanes, wait_current=False)
        return imgs[-1]
    step.joins = True
    return step


def work_per_step(workload):
    """(gemm flops, attention flops, sampled logits bytes) of one bench step on one GPU"""
    cfg_name, B, T, L = WORKLOADS[workload]
    vq = ver2cfg["vit-s-vqgan"]
    pk = 3 * vq["enc"]["patch_size"] ** 2
    if cfg_name is None:
        ge, ae = vit_flops(vq["enc"], vq["embed_dim"], pk, True)
        gd, ad = vit_flops(vq["dec"], vq["embed_dim"], pk, False)
        return B * (ge + gd), B * (ae + ad), 0
    cfg = ver2cfg[cfg_name]
    gs, as_ = s2_step_flops(cfg, 1024, vq["embed_dim"], vq["n_embed"], L)
    gd, ad = vit_flops(vq["dec"], vq["embed_dim"], pk, False)
    return B * T * (gs + gd), B * T * (as_ + ad), B * T * 1024 * vq["n_embed"] * 4


def usable_cores():
    """threads this process may actually run on: affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(float(quota) / period)))
        except Exception:
            pass
    return max(1, min(n, 64))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(workload):
    """oracle/torch_port.py -- a functional port of the reference's torch-CPU path -- timed on this host's
    cores on a bounded sample of the same workload (fp32, B=1)."""
    from oracle import torch_port as TP
    cfg_name, B, T, L = WORKLOADS[workload]
    cores = usable_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline on {cores} threads (os.cpu_count()={os.cpu_count()})")
    torch.manual_seed(0)
    vq = ver2cfg["vit-s-vqgan"]
    with torch.no_grad():
        if cfg_name is None:
            m = pm.create_model(arch="vqgan", version="vit-s-vqgan", pretrained=False)
            p = {k: v.detach() for k, v in m.state_dict().items()}
            x = torch.rand(1, 3, 256, 256) * 2 - 1
            TP.vqgan_decode(TP.vqgan_encode(x, p, vq)[0], p, vq)            # warm-up
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                TP.vqgan_decode(TP.vqgan_encode(x, p, vq)[0], p, vq)
            dt = (time.perf_counter() - t0) / reps
            return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
                    "sample": f"{reps} x (encode+decode) of one 256x256 image, torch-CPU fp32 port of the reference, B=1"}
        pipe = Pipeline(pm.Config(ver2cfg[cfg_name]), stage1_pretrained=False)
        p = {k: v.detach() for k, v in pipe.state_dict().items() if not k.startswith("text_model")}
        ids = torch.full((1, 1024), vq["n_embed"], dtype=torch.long)
        ctx = None if L is None else torch.randn(1, L, ver2cfg[cfg_name]["context_dim"])
        TP.sample_step(ids, 0.9, ctx, 5, 1.0, torch.rand(1, 1024, vq["n_embed"]), p, vq, ver2cfg[cfg_name])   # warm-up
        t0 = time.perf_counter()
        for step in range(T):
            noise = torch.rand(1, 1024, vq["n_embed"])
            ids, _, _ = TP.sample_step(ids, TP.mask_schedule((step + 1) / T), ctx, 5, 1.0 * (1 - step / T), noise, p, vq,
                                       ver2cfg[cfg_name])
        dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"one full {T}-step generate of ONE image (B=1, every step with its ViT decode), torch-CPU fp32 port "
                      f"of the reference on {cores} threads"}


def pmc_traffic(workload, dtype):
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json: FETCH_SIZE and
    WRITE_SIZE collected in separate --pmc runs of this same command, read side doubled per the gfx950 correction).
    Only valid for the configuration it was collected on."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if workload != "maskgit-uncond-12L-d512-T8" or dtype != "bf16" or not os.path.exists(path):
        return None
    try:
        fam = json.load(open(path))["families"]["gemm"]
        return {"hbm_bytes_per_launch": round(fam["hbm_bytes_per_launch"]), "read": round(fam["hbm_read_bytes_per_launch"]),
                "write": round(fam["hbm_write_bytes_per_launch"]), "source": "profiles/r01_pmc_traffic.json"}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="maskgit-uncond-12L-d512-T8", choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--final-decode-only", action="store_true", help="decode only the last step's image (not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs a {args.gpus}-rank launch (torch.distributed.run), WORLD_SIZE={world}")
    if rank != 0:
        # only rank 0 reports; the other ranks' stdout would only carry library banners (RCCL prints its version
        # there) that could land after rank 0's JSON line
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or os.environ.get("PM_BENCH_FORCE_DIST") == "1":      # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    log(f"building {args.workload} ({args.dtype}) on rank {rank}/{world}")
    model, _ = build(args.workload, device, dtype)
    step = make_step(args.workload, model, device, rank, decode_every_step=not args.final_decode_only)
    B = WORKLOADS[args.workload][1]

    recv = {}                                    # rank 0: receive buffers per (lane, shape), allocated once

    def gather(last, lane=0):
        """the path's only collective: finished images -> rank 0 (RCCL gather on the calling stream)"""
        if dist is None:
            return last
        bufs = None
        if rank == 0:
            key = (lane, tuple(last.shape))
            if key not in recv:
                recv[key] = [torch.empty_like(last) for _ in range(world)]
            bufs = recv[key]
        dist.gather(last, bufs, dst=0)
        return last

    def gather_lanes(parts):
        """free-running lanes: every lane hands its finished images to the gather on ITS stream, in lane order on all
        ranks; the lane's next replay is ordered after its gather by the stream, nothing waits on the host"""
        for lane, (_, imgs, st) in enumerate(parts):
            with torch.cuda.stream(st):
                gather(imgs[-1], lane)

    # one-time setup, not a benchmark step: the first two calls size the workspaces and capture the decode-loop
    # graph of every lane (eager pass, capture pass); afterwards every call is a pure replay
    log("setup (workspace sizing + hipGraph capture)")
    for i in range(2):
        step(-1 - i)
    torch.cuda.synchronize(device)
    log("warm-up")
    free_running = getattr(step, "joins", False) and STREAMS > 1
    for i in range(args.warmup):
        if free_running and dist is not None:
            gather_lanes(step(i, join=False))           # same code path as the timed loop (RCCL init, receive buffers)
        else:
            gather(step(i))
    log("timed region")
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(args.steps):
        if free_running:
            parts = step(args.warmup + i, join=False)   # lanes keep their own stream order; device-wide sync below joins them
            if dist is not None:
                gather_lanes(parts)
        else:
            gather(step(args.warmup + i))
    torch.cuda.synchronize(device)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed

    result = {
        "metric": "256x256 images/sec, 8-step MaskGIT decode (vit-s-vqgan), whole job", "value": round(value, 3),
        "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic (seeded random-init weights, all-masked start ids, Philox sampling noise)",
        "config": {"workload": args.workload, "batch_per_gpu": B, "timesteps": WORKLOADS[args.workload][2],
                   "topk": 5, "decode": "final step only" if args.final_decode_only else "every step (reference-equivalent work)",
                   "parallelism": f"dp{world} (independent images, no data-path collective)",
                   "hip_graph": bool(USE_GRAPH), "concurrent_micro_batches": STREAMS},
        "images_per_s_per_gpu": round(value / world, 3),
    }

    log(f"timed region done: {ms_per_step:.1f} ms/step")
    if rank == 0 and not args.no_roofline:
        # per-family kernel time of ONE more step, bracketed by hipEvents on the launch stream
        gf, af, sample_bytes = work_per_step(args.workload)
        if args.final_decode_only and WORKLOADS[args.workload][0] is not None:
            T = WORKLOADS[args.workload][2]
            vq = ver2cfg["vit-s-vqgan"]
            gd, ad = vit_flops(vq["dec"], vq["embed_dim"], 192, False)
            gf -= B * (T - 1) * gd
            af -= B * (T - 1) * ad
        ops.timing_reset()
        ops.timing_enable(True)
        if WORKLOADS[args.workload][0] is None:
            step(10_000)
        else:
            step(10_000, streams=1)                    # timing on => one stream, eager loop, every launch bracketed
        torch.cuda.synchronize(device)
        ops.timing_enable(False)
        fam = {f: ops.timing_get(f) for f in ("gemm", "attention", "layernorm", "sample", "vq", "rowops")}
        n_g, ms_g = fam["gemm"]
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        ach = gf / (ms_g * 1e-3) / 1e12 if ms_g > 0 else 0.0
        result["roofline"] = {
            "kernel": "GEMM family (gemm256_kernel + gemm2b_kernel + gemm_nt_kernel, all launches of one step)", "bound": "mfma",
            "achieved": round(ach, 2), "peak": peak,
            "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": pmc_traffic(args.workload, args.dtype), "launches": n_g,
            "avg_launch_ms": round(ms_g / max(n_g, 1), 4), "algorithmic_gflop_per_launch": round(gf / max(n_g, 1) / 1e9, 2)}
        n_a, ms_a = fam["attention"]
        n_s, ms_s = fam["sample"]
        result["kernel_families"] = {
            f: {"launches": fam[f][0], "ms": round(fam[f][1], 3)} for f in fam}
        if ms_a > 0:
            result["kernel_families"]["attention"]["tflops"] = round(af / (ms_a * 1e-3) / 1e12, 2)
        if ms_s > 0 and sample_bytes:
            result["kernel_families"]["sample"]["logits_GBps"] = round(sample_bytes / (ms_s * 1e-3) / 1e9, 1)
        result["end_to_end_tflops_per_gpu"] = round((gf + af) / (ms_per_step * 1e-3) / 1e12, 2)
    if rank == 0 and world == 1 and WORKLOADS[args.workload][0] is not None and not args.final_decode_only:
        # secondary number (NOT the headline): same loop, but only the image of the last step is decoded
        alt = make_step(args.workload, model, device, rank, decode_every_step=False)
        for i in range(3):
            alt(i)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        for i in range(3):
            alt(3 + i, join=False) if STREAMS > 1 else alt(3 + i)
        torch.cuda.synchronize(device)
        result["extra"] = {"final_decode_only_images_per_s": round(3 * B / (time.perf_counter() - t1), 2)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.workload)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    # RCCL writes its version banner to the C stdout buffer; push that out first so the JSON line is the last line
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()
