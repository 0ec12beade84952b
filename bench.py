#!/usr/bin/env python3
"""Headline benchmark: 256x256 images/sec/GPU, 8-step MaskGIT decode with vit-s-vqgan tokens.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python bench.py --gpus N ...            (no WORLD_SIZE in the environment: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one full pass of the hot path over one batch of synthetic input that is already resident
in HBM: for the default workload (BASELINE.json configs[2]) that is ONE Pipeline.generate() of B=64
images -- 8 MaskGIT steps of the 12-layer d=512 transformer (context=None, so attn2 is a second
self-attention) plus a ViT decode of the sampled tokens after EVERY step, which is the work the
reference does per generate() (generate.py:165; nothing is skipped).  Images stay on the device.
Multi-GPU: weak scaling, every rank decodes its own 64 images (RNG keyed by global image index), no
collective in the data path, one RCCL gather of the finished images to rank 0 inside the timed region.

Before the timed region the exact timed configuration (bf16, hipGraph replay, concurrent micro-batch lanes)
is checked once against the eager single-stream loop: ids and every decoded image must be bit-identical,
otherwise no value is printed ("self_check").

One JSON line is printed by rank 0; see DESIGN.md for how `roofline` and `cpu_baseline` are derived.
"""
import argparse
import glob
import json
import os
import resource
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# A RANK of a multi-GPU job (WORLD_SIZE > 1) runs the loop eagerly under AMD_DIRECT_DISPATCH=0 (set HERE, before anything loads the
# HIP runtime): with direct dispatch the runtime's helper thread busy-polls for as long as graph work is outstanding -- a full core per
# rank, 8.2 of this pool's 16 cores for 8 ranks -- without it the helper sleeps (measured on one GPU, same box: 471.1 vs 470.9 images/s,
# 0.12 vs 1.02 cores per rank; profiles/r05_c_host_polling.txt).  hipGraph replay is NOT usable in that runtime mode on ROCm 7.2
# (tools/hwtests/graph_dispatch_mode.hip: 39 of 40 replays of a chain of dependent kernels wrong, with no code of this repository
# involved), so the rank loop is the eager one -- bit-identical results, and at B = 64 per rank its launches hide behind the kernels.
# PM_BENCH_RANK_GRAPH=1 keeps direct dispatch + graph replay in ranks.  A single-GPU run (the headline) is unchanged: graph replay.
# PM_BENCH_FORCE_DIST=1 (one GPU, one RCCL rank) runs exactly what a rank runs: the way to time the rank mode without a node.
_RANK_MODE = ((int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("PM_BENCH_FORCE_DIST") == "1")
              and os.environ.get("PM_BENCH_RANK_GRAPH", "0") != "1")
if _RANK_MODE:
    os.environ.setdefault("AMD_DIRECT_DISPATCH", "0")
USE_GRAPH = os.environ.get("PM_BENCH_NO_GRAPH", "0") != "1" and os.environ.get("AMD_DIRECT_DISPATCH", "1") != "0"   # decode loop = replayed hipGraphs (captured during warm-up)
STREAMS = int(os.environ.get("PM_BENCH_STREAMS", "2"))   # concurrent micro-batches per GPU (1 = one stream); 2 measured best
LANE_SPLIT = os.environ.get("PM_BENCH_LANE_SPLIT")       # development: explicit micro-batch sizes, e.g. "32,16,16"
GATHER_MODE = os.environ.get("PM_BENCH_GATHER_MODE", "lane")   # lane | side | async: how a free-running lane issues its gather (development A/B)
PACE = int(os.environ.get("PM_BENCH_PACE", "2"))        # steps the host may run ahead of the GPU in the timed loop
PEAK_BF16_TFLOPS = 2500.0     # dense MFMA bf16, MI355X_MICROARCH.md chip table
PEAK_F32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0

DEFAULT_WORKLOAD = "maskgit-uncond-12L-d512-T8"
WORKLOADS = {
    # name: (pipeline config, batch per GPU, timesteps, context length or None)
    "maskgit-uncond-12L-d512-T8": ("bench-uncond-12L-d512", 64, 8, None),      # BASELINE configs[2]
    "maskgit-text-24L-d768-T8": ("bench-text-24L-d768", 32, 8, 77),            # north_star target model, 8 steps
    "maskgit-text-24L-d768-T12": ("bench-text-24L-d768", 32, 12, 77),          # BASELINE configs[3] per-GPU share
    "maskgit-text-24L-d1024-512px-T18": ("bench-text-24L-d1024-512px", 64, 18, 77),   # BASELINE configs[4] per-GPU share
    # the reference's ONLY pipeline preset (config.py:70-82, the default of factory.py:6): 12L / d1024 / 16 heads, T5-L features of
    # width 1024 (context_proj = Identity), at the reference's default 18 steps; the T5 tower is replaced by synthetic features
    "paintmindv1-T18": ("paintmindv1", 32, 18, 77),
    # BASELINE configs[3] as a user of a Muse-style model would run it: guided (GUIDANCE below), two tower passes per step
    "maskgit-text-24L-d768-T12-cfg3": ("bench-text-24L-d768", 32, 12, 77),
    "vit-s-recon": (None, 64, 0, None),                                         # BASELINE configs[1]
    "launch-selftest": (None, 4, 0, None),     # no compute: exercises the rank launcher / gather / JSON relay on CPU (gloo)
}
# guidance scale of a workload (absent = the reference's unguided step): logits = uncond + scale * (cond - uncond), the native loop of
# pmhip_pipeline_generate_guided (graph-captured, lanes)
GUIDANCE = {"maskgit-text-24L-d768-T12-cfg3": 3.0}
# measured after the headline, outside its timed region, each with its own ms_per_step ("extra" in the JSON line)
EXTRA_WORKLOADS = [("maskgit-text-24L-d768-T8", "bf16", 3), ("maskgit-text-24L-d768-T12", "bf16", 3), ("vit-s-recon", "bf16", 5),
                   ("maskgit-text-24L-d1024-512px-T18", "bf16", 2), (DEFAULT_WORKLOAD, "fp32", 2),
                   ("maskgit-text-24L-d768-T8", "fp32", 1), ("paintmindv1-T18", "bf16", 2), ("maskgit-text-24L-d768-T12-cfg3", "bf16", 2)]      # the north_star model in the mode parity is graded in (target >= 20 images/s)


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


# ---------------------------------------------------------------------------------------------
# rank launcher: `python bench.py --gpus N` with no WORLD_SIZE starts N fresh child processes (one per GPU) and
# relays rank 0's JSON line.  Decided before anything in this process touches the GPU; the parent never does.
# ---------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n):
    port = str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        out = subprocess.PIPE if r == 0 else subprocess.DEVNULL
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out))
    text, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [ln for ln in text.decode(errors="replace").splitlines() if ln.strip().startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    rc = max(abs(c) for c in codes)
    if rc == 0 and not lines:
        rc = 1
    return rc


# ---------------------------------------------------------------------------------------------
# algorithmic work (2*M*N*K per GEMM, 4*N*Nkv*inner per attention core)
# ---------------------------------------------------------------------------------------------
def layer_flops(D, heads, mlp_dim, N, stage2, ctx_len, kinds=None):
    """(GEMM flops, attention-core flops) of one transformer block; `kinds` (optional dict) accumulates the GEMM flops and the
    algorithmic HBM bytes of the residual producers per kernel kind (the timing families of include/pmhip.h)"""
    from paintmind_amd.ops import swiglu_hidden
    inner, hf = heads * 64, swiglu_hidden(mlp_dim)
    qkv, outp = 3 * 2 * N * D * inner, 2 * N * D * inner
    proj_self = qkv + outp
    core_self = 4 * N * N * inner
    gemm, attn = proj_self, core_self
    n_out = 1
    if stage2:
        n_out = 2
        if ctx_len is None:
            gemm += proj_self
            attn += core_self
            qkv *= 2
        else:
            gemm += 2 * 2 * N * D * inner + 2 * 2 * ctx_len * D * inner
            attn += 4 * N * ctx_len * inner
            qkv += 2 * N * D * inner                 # the query projection; the context's K / V are computed once per loop
    gemm += 6 * N * D * hf
    if kinds is not None:
        kinds["gemm_heads"] = kinds.get("gemm_heads", 0) + qkv
        kinds["gemm_resid2b"] = kinds.get("gemm_resid2b", 0) + n_out * outp
        kinds["gemm_swiglu"] = kinds.get("gemm_swiglu", 0) + 4 * N * D * hf
        kinds["gemm_resid"] = kinds.get("gemm_resid", 0) + 2 * N * D * hf
        # residual producers: A operand (bf16) + the residual pair read + written (4 + 4 bytes per element)
        kinds["bytes_resid2b"] = kinds.get("bytes_resid2b", 0) + n_out * N * (inner * 2 + D * 8)
        kinds["bytes_resid"] = kinds.get("bytes_resid", 0) + N * (hf * 2 + D * 8)
    return gemm, attn


def _scaled(kinds, into, times):
    for k, v in kinds.items():
        into[k] = into.get(k, 0) + v * times


def vit_flops(tower, embed_dim, patch_k, encode, n_embed=8192, kinds=None, times=1):
    N = (tower["image_size"] // tower["patch_size"]) ** 2
    lk = {}
    g, a = layer_flops(tower["dim"], tower["num_head"], tower["mlp_dim"], N, False, None, lk)
    gemm, attn = g * tower["depth"], a * tower["depth"]
    gemm += 2 * N * patch_k * tower["dim"] + 2 * N * tower["dim"] * embed_dim
    if encode:
        gemm += 2 * N * embed_dim * n_embed       # the VQ distance product (quantize.py:26)
    if kinds is not None:
        _scaled(lk, kinds, tower["depth"] * times)
        if encode:                                # patch embedding + prev_quant: plain
            kinds["gemm_plain"] = kinds.get("gemm_plain", 0) + times * (2 * N * patch_k * tower["dim"] + 2 * N * tower["dim"] * embed_dim)
        else:                                     # post_quant opens the residual stream, the pixel projection is plain
            kinds["gemm_resid"] = kinds.get("gemm_resid", 0) + times * 2 * N * tower["dim"] * embed_dim
            kinds["gemm_plain"] = kinds.get("gemm_plain", 0) + times * 2 * N * patch_k * tower["dim"]
    return gemm, attn


def s2_step_flops(cfg, N, embed_dim, n_embed, ctx_len, kinds=None, times=1, loops=0):
    """`times` = images x steps; `loops` = images: what this build computes once per decode loop and the reference every step
    (context_proj and the cross-attention K / V of the static context, stage2/transformer.py:84-85, attention.py:48-49) is
    counted per step in the totals (algorithmic work) and once per loop in `kinds` (what the kernels execute)"""
    lk = {}
    g, a = layer_flops(cfg["dim"], cfg["num_head"], cfg["mlp_dim"], N, True, ctx_len, lk)
    gemm, attn = g * cfg["depth"], a * cfg["depth"]
    gemm += 2 * N * embed_dim * cfg["dim"] + 2 * N * cfg["dim"] * n_embed
    if kinds is not None:
        _scaled(lk, kinds, cfg["depth"] * times)
        kinds["gemm_resid"] = kinds.get("gemm_resid", 0) + times * 2 * N * embed_dim * cfg["dim"]     # token_proj
        kinds["gemm_plain"] = kinds.get("gemm_plain", 0) + times * 2 * N * cfg["dim"] * n_embed       # to_logits
        if ctx_len is not None:
            inner = cfg["num_head"] * 64
            kinds["gemm_heads"] += loops * cfg["depth"] * 2 * 2 * ctx_len * cfg["dim"] * inner
            cdim = cfg.get("context_dim") or {"t5-l": 1024, "t5-xl": 2048}[cfg["t5"]]
            if cdim != cfg["dim"]:
                kinds["gemm_plain"] += loops * 2 * ctx_len * cdim * cfg["dim"]
    return gemm, attn


def stage1_cfg(cfg_name):
    from paintmind_amd.config import ver2cfg
    return ver2cfg["vit-s-vqgan"] if cfg_name is None else ver2cfg[ver2cfg[cfg_name]["stage1"]]


def work_per_step(workload, decode_every_step=True, kinds=None):
    """(gemm flops, attention flops, sampled logits bytes) of one bench step on one GPU; `kinds` (optional dict) receives the
    GEMM flops / residual-producer bytes per kernel kind"""
    from paintmind_amd.config import ver2cfg
    cfg_name, B, T, L = WORKLOADS[workload]
    vq = stage1_cfg(cfg_name)
    pk = 3 * vq["enc"]["patch_size"] ** 2
    if cfg_name is None:
        ge, ae = vit_flops(vq["enc"], vq["embed_dim"], pk, True, vq["n_embed"], kinds, B)
        gd, ad = vit_flops(vq["dec"], vq["embed_dim"], pk, False, kinds=kinds, times=B)
        return B * (ge + gd), B * (ae + ad), 0
    cfg = ver2cfg[cfg_name]
    N = (vq["enc"]["image_size"] // vq["enc"]["patch_size"]) ** 2
    n_dec = T if decode_every_step else 1
    gs, as_ = s2_step_flops(cfg, N, vq["embed_dim"], vq["n_embed"], L, kinds, B * T, B)
    if workload in GUIDANCE:                                  # + the unconditional tower pass of every guided step
        gu, au = s2_step_flops(cfg, N, vq["embed_dim"], vq["n_embed"], None, kinds, B * T, B)
        gs, as_ = gs + gu, as_ + au
    gd, ad = vit_flops(vq["dec"], vq["embed_dim"], pk, False, kinds=kinds, times=B * n_dec)
    # bytes the sampling kernel reads per row: the block statistics (8 B per 64 columns) + the top-k blocks (256 B each) where the
    # tile sampler runs (top-k <= 8, V % 64 == 0: sample.hip), else the whole fp32 row
    V, k = vq["n_embed"], 5
    tiles = V % 64 == 0 and k <= 8
    return B * (T * gs + n_dec * gd), B * (T * as_ + n_dec * ad), B * T * N * ((V // 64) * 8 + k * 256 if tiles else V * 4)


# ---------------------------------------------------------------------------------------------
def build(workload, device, dtype):
    import torch
    import paintmind_amd as pm
    from paintmind_amd.config import ver2cfg
    from paintmind_amd.generate import Pipeline
    cfg_name = WORKLOADS[workload][0]
    torch.manual_seed(0)
    if cfg_name is None:
        model = pm.create_model(arch="vqgan", version="vit-s-vqgan", pretrained=False).to(device).eval()
        model.set_compute_dtype(dtype)
        return model
    text_model = None
    if "text_model" not in ver2cfg[cfg_name]:               # a reference preset: its T5 tower is a download, stand in for it
        from paintmind_amd.modules.encoder import SyntheticTextEmbedder
        text_model = SyntheticTextEmbedder(context_dim_of(cfg_name))
    pipe = Pipeline(pm.Config(ver2cfg[cfg_name]), stage1_pretrained=False, text_model=text_model).to(device).eval()
    pipe.set_compute_dtype(dtype)
    return pipe


def context_dim_of(cfg_name):
    from paintmind_amd.config import ver2cfg
    from paintmind_amd.generate import T5_TXT_DIM
    cfg = ver2cfg[cfg_name]
    return cfg.get("context_dim") or T5_TXT_DIM[cfg["t5"]]


def lanes_arg(streams=None):
    s = STREAMS if streams is None else streams
    if LANE_SPLIT and s > 1:
        return tuple(int(x) for x in LANE_SPLIT.split(","))
    return s


def make_step(workload, model, device, rank, decode_every_step=True):
    import torch
    from paintmind_amd.config import ver2cfg
    cfg_name, B, T, L = WORKLOADS[workload]
    if cfg_name is None:
        x = (torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(rank)) * 2 - 1).to(device)

        def step(i):
            z, _, _ = model.encode(x)
            return model.decode(z)
        step.inputs = x
        return step
    pipe = model
    ctx = None
    if L is not None:
        g = torch.Generator().manual_seed(1234 + rank)
        ctx = torch.randn(B, L, context_dim_of(cfg_name), generator=g).to(device)
    flags = [True] * T if decode_every_step else [t == T - 1 for t in range(T)]
    guidance = GUIDANCE.get(workload)

    def step(i, join=True, streams=None):
        s = STREAMS if streams is None else streams
        # join=False (single-GPU timed loop): the micro-batch lanes are not joined between steps, so consecutive
        # steps pipeline across lanes; the caller joins once before the closing synchronize
        if s > 1 and not join:
            return pipe.generate_ids(ctx, B, T, 1.0, 5, flags, seed=1000 + i, image_base=rank * B, use_graph=USE_GRAPH,
                                     streams=lanes_arg(s), join=False, wait_current=False, guidance_scale=guidance)
        ids, imgs = pipe.generate_ids(ctx, B, T, 1.0, 5, flags, seed=1000 + i, image_base=rank * B, use_graph=USE_GRAPH,
                                      streams=lanes_arg(s), wait_current=False, guidance_scale=guidance)
        return imgs[-1]
    step.joins = True
    step.ctx, step.flags = ctx, flags
    return step


def self_check(workload, model, step, device, rank):
    """The timed configuration (compute dtype as built, hipGraph replay, concurrent lanes) against the eager single-stream
    loop on the same seed: ids and every decoded image bit-identical (reference generate.py:183-198 has ONE code path).
    Runs after the setup calls (graphs captured), outside the timed region.  Returns (ok, detail)."""
    import torch
    cfg_name, B, T, L = WORKLOADS[workload]
    if cfg_name is None:
        x = step.inputs
        z, _, idx = model.encode(x)
        rec, rec2 = model.decode(z), model.decode_from_indice(idx)
        z1, _, idx1 = model.encode(x[:2].contiguous())
        ok = bool(torch.isfinite(rec).all()) and float(rec.abs().max()) <= 1.0 and torch.equal(idx[:2], idx1) \
            and float((rec - rec2).abs().max()) < 2e-2
        detail = "encode/decode finite, clamped, batch-invariant tokens, decode(z) == decode_from_indice(idx)"
        if model.compute_dtype == torch.bfloat16:
            # the timed bf16 path against fp32-verify (the mode parity with the reference is graded in) ON THE TIMED INPUT:
            # token agreement and reconstruction deviation must sit inside the gates of tests/test_gpu_model.py
            stats = recon_bf16_vs_fp32(model, x, idx, rec)
            ok = ok and stats["token_agreement"] >= RECON_GATE["token_agreement_min"] and \
                stats["rec_mean_abs_dev"] <= RECON_GATE["rec_mean_abs_dev_max"] and stats["rec_max_abs_dev"] <= RECON_GATE["rec_max_abs_dev_max"]
            detail += f"; bf16 vs fp32-verify on the timed input: {stats}, gate {RECON_GATE}"
            self_check.last_recon_stats = stats
        return ok, detail
    seed = 424242
    kw = dict(seed=seed, image_base=rank * B, guidance_scale=GUIDANCE.get(workload))
    ids_t, imgs_t = model.generate_ids(step.ctx, B, T, 1.0, 5, step.flags, use_graph=USE_GRAPH, streams=lanes_arg(), **kw)
    ids_e, imgs_e = model.generate_ids(step.ctx, B, T, 1.0, 5, step.flags, use_graph=False, streams=1, **kw)
    torch.cuda.synchronize(device)
    ok = torch.equal(ids_t, ids_e) and torch.equal(imgs_t, imgs_e) and bool(torch.isfinite(imgs_e).all()) \
        and int((ids_e == model.mask_token_id).sum(1).max()) == 1 and int((ids_e == model.mask_token_id).sum(1).min()) == 1
    return ok, (f"hipGraph={USE_GRAPH} lanes={lanes_arg()} ids+images bit-identical to the eager single-stream loop, "
                f"one residual mask token per image")


# gates of the bf16 ViT path against fp32-verify at B = 64 (measured values and their spread: tests/test_gpu_model.py)
RECON_GATE = {"token_agreement_min": 0.9875, "rec_mean_abs_dev_max": 0.0036, "rec_max_abs_dev_max": 0.040}


def recon_bf16_vs_fp32(model, x, idx16, rec16_own, chunk=8):
    """fp32-verify encode/decode of the timed input in chunks (the exact-f32 matrix path is ~15x slower: checker only) against
    the bf16 results: token agreement, deviation of the bf16 decode of the SAME (fp32) latent, deviation end to end"""
    import torch
    model.set_compute_dtype(torch.float32)
    try:
        z32, idx32, rec32 = [], [], []
        for i in range(0, x.shape[0], chunk):
            z, _, idx = model.encode(x[i:i + chunk])
            z32.append(z); idx32.append(idx); rec32.append(model.decode(z))
        z32, idx32, rec32 = torch.cat(z32), torch.cat(idx32), torch.cat(rec32)
    finally:
        model.set_compute_dtype(torch.bfloat16)
    d = (model.decode(z32) - rec32).abs()
    return {"token_agreement": round(float((idx16 == idx32).float().mean()), 5), "rec_mean_abs_dev": round(float(d.mean()), 5),
            "rec_max_abs_dev": round(float(d.max()), 4), "rec_mean_abs_dev_own_tokens": round(float((rec16_own - rec32).abs().mean()), 5)}


def time_steps(step, device, n_setup, n_warm, n_timed, free_running):
    import torch
    for i in range(n_setup):
        step(-1 - i)
    torch.cuda.synchronize(device)
    for i in range(n_warm):
        step(i)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for i in range(n_timed):
        if free_running:
            step(n_warm + i, join=False)
        else:
            step(n_warm + i)
    torch.cuda.synchronize(device)
    return (time.perf_counter() - t0) / n_timed


def extra_workload(name, dtype_name, steps, device):
    """one more workload under the same harness (setup 2 = sizing + graph capture, 1 warm-up, `steps` timed)"""
    import torch
    dtype = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    model = build(name, device, dtype)
    step = make_step(name, model, device, 0)
    pipeline = WORKLOADS[name][0] is not None
    from paintmind_amd import ops
    ops.attention_fallbacks(reset=True, device=device)
    dt = time_steps(step, device, 2, 1, steps, pipeline and STREAMS > 1)
    fallbacks = ops.attention_fallbacks(reset=True, device=device) if dtype_name == "bf16" else None
    ok, _ = self_check(name, model, step, device, 0)
    B, T = WORKLOADS[name][1], WORKLOADS[name][2]
    gf, af, _ = work_per_step(name)
    out = {"images_per_s": round(B / dt, 2) if ok else None, "ms_per_step": round(dt * 1e3, 3) if ok else None, "batch": B,
           "timesteps": T, "steps": steps, "dtype": dtype_name, "tflops": round((gf + af) / dt / 1e12, 1) if ok else None,
           "self_check": "ok" if ok else "FAILED (no value is reported for a configuration that fails its check)",
           # workgroups of the bf16 attention kernel re-run through its exact path (setup + warm-up + timed steps)
           "attention_fallbacks": fallbacks}
    if not pipeline and dtype_name == "bf16":
        out["bf16_vs_fp32_verify"] = getattr(self_check, "last_recon_stats", None)
        out["gate"] = RECON_GATE
    del step, model
    torch.cuda.empty_cache()
    return out


def dropin_generate(workload, model, device, steps):
    """The reference's call, nothing opted into: Pipeline.generate(text, timesteps, temperature, topk, save_interval)
    (generate.py:183-198) -> list of CPU tensors.  Text tower (synthetic stand-in, or none for the unconditional workload)
    and the device-to-host copies of the returned images are inside the timed calls."""
    import torch
    from paintmind_amd.modules.encoder import NullTextEmbedder
    cfg_name, B, T, L = WORKLOADS[workload]
    text = [f"prompt {i}" for i in range(B)]
    if L is None:
        model.text_model = NullTextEmbedder()
    out = {}
    for si in (1, 2):
        kw = dict(timesteps=T, temperature=1.0, topk=5, save_interval=si)
        for i in range(3):                                   # eager pass, capture pass, first replay
            model.generate(text, seed=1 + i, **kw)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(steps):
            imgs = model.generate(text, seed=100 + i, **kw)
        dt = (time.perf_counter() - t0) / steps              # generate() returns with every image on the host
        out[f"save_interval_{si}"] = {"images_per_s": round(B / dt, 2), "ms_per_call": round(dt * 1e3, 3), "returned_images": len(imgs),
                                      "d2h_MB_per_call": round(sum(x.numel() for x in imgs) * 4 / 1e6, 1),
                                      "host_tensors_pinned": bool(imgs[0].is_pinned())}
    return out


def small_batch_latency(workload, model, device):
    """B = 1, 2 and 8 through the same decode loop (graph replay, one stream).  Round 5: small launches take small-batch forms of
    the same arithmetic (64 / 128 queries per attention workgroup, the folded LayerNorm on the four-stage 128x128 GEMM with the
    fold coefficients computed in its prologue), bit-identical to the large-batch kernels, so results stay batch-invariant"""
    import torch
    cfg_name, _, T, L = WORKLOADS[workload]
    out = {}
    for Bs in (1, 2, 8):
        ctx = None if L is None else torch.randn(Bs, L, model.transformer.context_dim if hasattr(model.transformer, "context_dim") else 768).to(device)
        flags = [True] * T
        for i in range(3):                                   # eager pass, capture pass, first replay
            model.generate_ids(ctx, Bs, T, 1.0, 5, flags, seed=1 + i, use_graph=USE_GRAPH, streams=1)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for i in range(5):
            model.generate_ids(ctx, Bs, T, 1.0, 5, flags, seed=10 + i, use_graph=USE_GRAPH, streams=1)
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / 5
        out[f"B{Bs}"] = {"ms_per_generate": round(dt * 1e3, 2), "images_per_s": round(Bs / dt, 2)}
    return out


def usable_cores():
    """threads this process may actually run on: affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    limited = False
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(float(quota) / period))
                if q < n:
                    n, limited = q, True
        except Exception:
            pass
    return max(1, min(n, 64)), limited


def cpu_baseline(workload):
    """oracle/torch_port.py -- a functional port of the reference's torch-CPU path -- timed on this host's
    cores on a bounded sample of the same workload (fp32): B=1 best of 3, and B=8 on two of the T steps."""
    import torch
    import paintmind_amd as pm
    from oracle import torch_port as TP
    from paintmind_amd.config import ver2cfg
    from paintmind_amd.generate import Pipeline
    cfg_name, B, T, L = WORKLOADS[workload]
    cores, limited = usable_cores()
    torch.set_num_threads(cores)
    host = os.cpu_count()
    cores_note = (f"{cores} threads = this container's cgroup CPU quota; the host has {host} logical cores"
                  if limited else f"{cores} threads (affinity mask; host has {host} logical cores)")
    log(f"cpu_baseline on {cores_note}")
    torch.manual_seed(0)
    vq = stage1_cfg(cfg_name)
    base = {"unit": "images/s", "cores": cores, "cores_note": cores_note, "host_cores": host, "kind": "port"}
    with torch.no_grad():
        if cfg_name is None:
            m = pm.create_model(arch="vqgan", version="vit-s-vqgan", pretrained=False)
            p = {k: v.detach() for k, v in m.state_dict().items()}
            x = torch.rand(1, 3, 256, 256) * 2 - 1
            TP.vqgan_decode(TP.vqgan_encode(x, p, vq)[0], p, vq)            # warm-up
            best = 1e30
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(4):
                    TP.vqgan_decode(TP.vqgan_encode(x, p, vq)[0], p, vq)
                best = min(best, (time.perf_counter() - t0) / 4)
            x8 = torch.rand(8, 3, 256, 256) * 2 - 1
            t0 = time.perf_counter()
            TP.vqgan_decode(TP.vqgan_encode(x8, p, vq)[0], p, vq)
            b8 = 8 / (time.perf_counter() - t0)
            return dict(base, value=round(1.0 / best, 4), value_b8=round(b8, 4),
                        sample="best of 3 x (4 x encode+decode of one 256x256 image), torch-CPU fp32 port of the reference, B=1; "
                               "value_b8: one encode+decode at B=8")
        pipe = Pipeline(pm.Config(ver2cfg[cfg_name]), stage1_pretrained=False)
        p = {k: v.detach() for k, v in pipe.state_dict().items() if not k.startswith("text_model")}
        N = pipe.num_tokens

        def run(Bc, steps):
            ids = torch.full((Bc, N), vq["n_embed"], dtype=torch.long)
            ctx = None if L is None else torch.randn(Bc, L, context_dim_of(cfg_name))
            t0 = time.perf_counter()
            for step in steps:
                noise = torch.rand(Bc, N, vq["n_embed"])
                ids, _, _ = TP.sample_step(ids, TP.mask_schedule((step + 1) / T), ctx, 5, 1.0 * (1 - step / T), noise, p, vq,
                                           ver2cfg[cfg_name])
            return time.perf_counter() - t0
        run(1, [0])                                                         # warm-up
        best = min(run(1, range(T)) for _ in range(3))
        t8 = run(8, [0, 1])                                                 # every step is a full forward + decode: same cost
    return dict(base, value=round(1.0 / best, 4), value_b8_extrapolated=round(8.0 / (t8 * T / 2), 4),
                sample=f"best of 3 full {T}-step generates of ONE image (B=1, every step with its ViT decode), torch-CPU fp32 port "
                       f"of the reference; value_b8_extrapolated: B=8 timed on 2 of the {T} steps (all steps cost the same) and "
                       f"scaled to {T}")


def pmc_traffic(workload, dtype, family):
    """HBM bytes per launch of a kernel family from the newest committed rocprofv3 PMC passes (profiles/r*_pmc_traffic.json:
    FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs of this same command by tools/profile_round.sh, read side
    doubled per the gfx950 correction).  NOT measured in this run; only valid for the configuration it was collected on."""
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if workload != DEFAULT_WORKLOAD or dtype != "bf16" or not paths:
        return None
    try:
        fam = json.load(open(paths[-1]))["families"][family]
        return {"hbm_bytes_per_launch": round(fam["hbm_bytes_per_launch"]), "read": round(fam["hbm_read_bytes_per_launch"]),
                "write": round(fam["hbm_write_bytes_per_launch"]), "source": os.path.relpath(paths[-1], ROOT),
                "note": "from the committed profile of this command (tools/profile_round.sh), not collected in this run"}
    except Exception:
        return None


CALIBRATION_REFERENCE_TFLOPS = 1500.0     # the 8192^3 rate `frac_calibrated` is normalised to (middle of the pool's 1.40-1.65 PFLOP/s)


def gemm_calibration(device, iters=8):
    """this library's 8192 x 8192 x 8192 bf16 GEMM on this box, hipEvent-timed (about 10 ms in total)"""
    import torch
    from paintmind_amd import ops
    g = torch.Generator(device="cpu").manual_seed(5)
    a = (torch.rand(8192, 8192, generator=g) * 2 - 1).to(device, torch.bfloat16)
    w = (torch.rand(8192, 8192, generator=g) * 2 - 1).to(device, torch.bfloat16)
    for _ in range(3):
        ops.gemm(a, w, out_dtype=torch.bfloat16)
    best = 1e30
    for _ in range(3):                                       # best of three batches: the rate drifts with the chip's power state
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.gemm(a, w, out_dtype=torch.bfloat16)
        e1.record()
        torch.cuda.synchronize(device)
        best = min(best, e0.elapsed_time(e1) / iters)
    ms = best
    return {"gemm_8192_bf16_tflops": round(2 * 8192 ** 3 / (ms * 1e-3) / 1e12, 1), "reference_tflops": CALIBRATION_REFERENCE_TFLOPS,
            "note": "frac_calibrated = frac x reference / measured: a cross-box NORMALISER (pool spread +-3..8 %), not a roofline fraction -- the contract number is frac against the nominal peak"}


def thread_cpu_seconds():
    """{tid: (thread name, user + system CPU seconds)} of this process (Linux /proc)"""
    out = {}
    tck = os.sysconf("SC_CLK_TCK")
    try:
        for tid in os.listdir("/proc/self/task"):
            try:
                raw = open(f"/proc/self/task/{tid}/stat").read()
                name = raw[raw.index("(") + 1:raw.rindex(")")]
                f = raw[raw.rindex(")") + 2:].split()
                out[int(tid)] = (name, (int(f[11]) + int(f[12])) / tck)
            except Exception:
                pass
    except Exception:
        pass
    return out


class PowerSampler:
    """Best-effort package power / shader clock of this rank's GPU during the timed region (`rocm-smi` polled from a thread,
    about two samples per second; the subprocess waits outside the GIL).  The decode loop runs at the socket power cap
    (DESIGN.md section 4f): the line carries the evidence.  Never raises; `summary()` is None when nothing could be read."""

    def __init__(self, device_index, period=0.35):
        import threading
        self.dev, self.samples, self.stop, self.period = device_index, [], threading.Event(), period
        self.thread = threading.Thread(target=self._run, daemon=True)

    def _run(self):
        import re
        while not self.stop.is_set():
            try:
                # rocm-smi is a `#!/usr/bin/env python3` script: under rocprofv3 the profiler's preloaded tool library would
                # initialise the GPU inside `env`, whose exec of python3 the GPU boxes of this pool then refuse -- the child gets an
                # environment without the profiler's hooks
                env = {k: v for k, v in os.environ.items()
                       if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS"))}
                out = subprocess.run(["rocm-smi", "-d", str(self.dev), "--showpower", "--showclocks"], capture_output=True, text=True,
                                     timeout=3, env=env).stdout
                pw = re.search(r"Power \(W\): ([\d.]+)", out)
                ck = re.search(r"sclk clock level: \S+ \((\d+)Mhz", out)
                if pw and ck:
                    self.samples.append((float(pw.group(1)), float(ck.group(1))))
            except Exception:
                return
            self.stop.wait(self.period)

    def start(self):
        if os.environ.get("PM_BENCH_NO_POWER") != "1":
            self.thread.start()
        return self

    def summary(self):
        self.stop.set()
        if self.thread.is_alive():
            self.thread.join(timeout=4)
        s = self.samples[1:] if len(self.samples) > 2 else self.samples       # the first sample may predate the load
        if not s:
            return None
        return {"package_watts_mean": round(sum(x[0] for x in s) / len(s), 1), "package_watts_max": max(x[0] for x in s),
                "sclk_mhz_mean": round(sum(x[1] for x in s) / len(s)), "samples": len(s),
                "note": "rocm-smi during the timed region; the socket cap of this part is 1400 W, the nominal shader clock 2400 MHz"}


def kernel_clock_probe(device, seconds=1.2):
    """The shader clock and package power each kernel kind of the default workload settles at when it is looped ALONE at its bench
    shape for `seconds` (rocm-smi polled meanwhile): `frac` in roofline_by_kernel is quoted against the NOMINAL 2.4 GHz peak as the
    contract says; against the clock the part actually grants that kernel the same rate is frac * 2400 / sclk
    (`frac_at_measured_clock`, matrix-bound kernels only).  Returns {family: {...}}; never raises."""
    import torch
    out = {}
    try:
        from paintmind_amd import _lib, ops, packing
        lib = _lib.load()
        bf = torch.bfloat16
        M, D = 65536, 512
        g = torch.Generator(device="cpu").manual_seed(7)
        rn = lambda *shape: torch.randn(*shape, generator=g).to(device)
        hi, lo = ops.split_hilo(rn(M, D))
        coef = ops.ln_coef(hi)
        gamma, beta = torch.ones(D, device=device), torch.zeros(D, device=device)
        wg1, c1, d1 = packing.ln_fold(rn(1536, D) * D ** -0.5, gamma, beta)
        lin = torch.nn.Linear(D, 2 * 1368).to(device)
        w12p32, b12p, _ = packing.pack_w12(lin, torch.float32)
        wg2, c2, d2 = packing.ln_fold(w12p32, gamma, beta)
        q, k, vt = (rn(64, 8, 1024, 64) * 0.5).to(bf), rn(64, 8, 1024, 64).to(bf), rn(64, 8, 64, 1024).to(bf)
        a = (rn(M, D) * 0.7).to(bf)
        wo, bo = (rn(D, D) * D ** -0.5).to(bf), rn(D)
        parts = torch.empty(M, D // 64, 2, device=device)
        hid, w3 = (rn(M, 1408) * 0.5).to(bf), (rn(D, 1408) * 1408 ** -0.5).to(bf)
        wg3, c3, d3 = packing.ln_fold(rn(8192, D) * D ** -0.5, gamma, beta)
        bl = rn(8192)
        sp = ops.stream_ptr(device)
        hs = lambda A, K, W: lib.pmhip_gemm_hilo_stats(A.data_ptr(), K, W.data_ptr(), K, bo.data_ptr(), hi.data_ptr(), lo.data_ptr(), D, 0,
                                                       hi.data_ptr(), lo.data_ptr(), D, M, D, K, parts.data_ptr(), sp)
        kernels = {"attention": lambda: ops.attention(q, k, vt, 1024, use_exp2=True),
                   "gemm_heads": lambda: ops.gemm_heads_ln(hi, wg1, 8, 1024, [ops.PART_Q, ops.PART_K, ops.PART_V], 0.18, coef, c1, d1),
                   "gemm_swiglu": lambda: ops.gemm_swiglu_ln(hi, wg2, b12p, coef, c2, d2),
                   "gemm_resid": lambda: hs(hid, 1408, w3), "gemm_resid2b": lambda: hs(a, D, wo),
                   "gemm_plain": lambda: ops.gemm_ln(hi, wg3, coef, c3, d3, bias=bl, out_dtype=torch.float32)}
        for fam, fn in kernels.items():
            for _ in range(5):
                fn()
            torch.cuda.synchronize(device)
            ps = PowerSampler(device.index or 0, period=0.05).start()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0, n = time.time(), 0
            e0.record()
            while time.time() - t0 < seconds:
                for _ in range(40):
                    fn()
                n += 40
                torch.cuda.synchronize(device)
            e1.record()
            torch.cuda.synchronize(device)
            sm = ps.summary()
            if sm:
                out[fam] = {"sclk_mhz_alone": sm["sclk_mhz_mean"], "package_watts_alone": sm["package_watts_mean"],
                            "us_per_launch_alone": round(e0.elapsed_time(e1) / n * 1e3, 1), "samples": sm["samples"]}
    except Exception as e:
        out["error"] = f"{type(e).__name__}: {e}"
    return out


def blocking_sync(device_index):
    """PM_BENCH_BLOCKING_SYNC=1 (opt-in): hipDeviceScheduleBlockingSync before the device context exists, so that EVERY wait of
    the process sleeps on an interrupt instead of spinning.  Not the default: it also slows the host-paced drop-in generate()
    measured in `extra` (0.6 ms of wake-up latency per saved image: 411 vs 454 images/s).  The timed loop does not need it:
    its pacing waits are blocking events already, only the closing synchronize (<= PACE steps of work) spins."""
    if os.environ.get("PM_BENCH_BLOCKING_SYNC") != "1":
        return
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        hip.hipSetDevice(ctypes.c_int(device_index))
        hip.hipSetDeviceFlags(ctypes.c_uint(0x4))           # hipDeviceScheduleBlockingSync
    except Exception:
        pass


def flush_c_stdout():
    # RCCL writes its version banner to the C stdout buffer; push that out first so the JSON line is the last line
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def run_launch_selftest(args, rank, world):
    """no GPU, no compute: N gloo ranks, the shard / gather / max-over-ranks / JSON plumbing of the real run"""
    import torch
    import torch.distributed as dist
    from paintmind_amd.dist import gather_images
    if rank != 0:
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    B = WORKLOADS["launch-selftest"][1]
    t0 = time.perf_counter()
    for i in range(args.steps):
        local = torch.full((B, 3, 4, 4), float(rank))
        got = gather_images(local, [B] * world) if world > 1 else local
        if rank == 0:
            assert got.shape[0] == B * world and all(float(got[r * B, 0, 0, 0]) == r for r in range(world))
    elapsed = time.perf_counter() - t0
    per_rank = [elapsed]
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64)
        allt = [torch.zeros_like(tt) for _ in range(world)]
        dist.all_gather(allt, tt)
        per_rank = [float(x) for x in allt]
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "launch-selftest", "value": round(world * B * args.steps / max(per_rank), 3), "unit": "images/s",
                          "n_gpus": world, "ranks": world, "steps": args.steps, "warmup": args.warmup,
                          "per_rank_images_per_s": [round(B * args.steps / e, 3) for e in per_rank], "data": "none (launcher self-test)"}),
              flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=DEFAULT_WORKLOAD, choices=sorted(WORKLOADS))
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--final-decode-only", action="store_true", help="decode only the last step's image (not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads measured after the headline")
    args = ap.parse_args()

    force_dist = os.environ.get("PM_BENCH_FORCE_DIST") == "1"
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or force_dist):
        return launch_ranks(args.gpus)              # this process never touches the GPU

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.workload == "launch-selftest":
        return run_launch_selftest(args, rank, world)

    blocking_sync(local_rank)
    import torch
    from paintmind_amd import ops
    if rank != 0:
        # only rank 0 reports; the other ranks' stdout would only carry library banners (RCCL prints its version
        # there) that could land after rank 0's JSON line
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    dist = None
    if world > 1 or force_dist:                       # the env switch exercises the RCCL path on one GPU
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    log(f"building {args.workload} ({args.dtype}) on rank {rank}/{world}")
    model = build(args.workload, device, dtype)
    step = make_step(args.workload, model, device, rank, decode_every_step=not args.final_decode_only)
    B = WORKLOADS[args.workload][1]
    cfg0 = WORKLOADS[args.workload][0]
    img_px = stage1_cfg(cfg0)["enc"]["image_size"]

    recv = {}                                    # rank 0: receive buffers per (lane, shape), allocated once
    gather_streams = {}                          # lane -> the side stream its gathers are issued from
    inflight = []                                # tensors handed to a gather that has not been drained yet (kept alive)

    def gather(last, lane=0, asynchronous=False):
        """the path's only collective: finished images -> rank 0.  RCCL runs it on the process group's own stream, ordered
        after the calling stream's work so far.  `asynchronous` (the free-running lane loop): issued from the lane's own stream
        with the ordinary synchronous-API call, i.e. the LANE waits for the collective on the device (well under a millisecond
        per 134 ms step) and no host thread waits at all.  The two alternatives were measured with one RCCL rank in the rank
        mode above (profiles/r05_c_host_polling.txt, PM_BENCH_GATHER_MODE): `async_op=True` handles cost a thread spinning at a
        full core from the first use until the group is destroyed (1.12 cores per rank), a side stream per lane two (2.13);
        the plain call 0.12."""
        if dist is None:
            return last
        bufs = None
        if rank == 0:
            key = (lane, tuple(last.shape))
            if key not in recv:
                recv[key] = [torch.empty_like(last) for _ in range(world)]
            bufs = recv[key]
        mode = GATHER_MODE
        if asynchronous and mode == "async":
            inflight.append((dist.gather(last, bufs, dst=0, async_op=True), last))
        elif asynchronous and mode == "side":
            side = gather_streams.get(lane)
            if side is None:
                side = gather_streams[lane] = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                dist.gather(last, bufs, dst=0)
            last.record_stream(side)
            inflight.append(last)
        else:
            dist.gather(last, bufs, dst=0)
        return last

    def gather_lanes(parts):
        """free-running lanes: every lane issues the gather of its finished images from its own stream (in lane order on all
        ranks).  In the default mode that is the synchronous-API call: the LANE waits for the collective on the device (well
        under a millisecond per step), no host thread waits for RCCL inside the loop (see gather())"""
        for lane, (_, imgs, st) in enumerate(parts):
            with torch.cuda.stream(st):
                gather(imgs[-1], lane, asynchronous=True)

    def drain_gathers():
        for side in gather_streams.values():
            side.synchronize()
        for it in inflight:
            if isinstance(it, tuple):
                it[0].wait()
        inflight.clear()

    # one-time setup, not a benchmark step: the first two calls size the workspaces and capture the decode-loop
    # graph of every lane (eager pass, capture pass); afterwards every call is a pure replay
    log("setup (workspace sizing + hipGraph capture)")
    for i in range(2):
        step(-1 - i)
    torch.cuda.synchronize(device)
    log("self-check of the timed configuration against the eager single-stream loop")
    ok, detail = self_check(args.workload, model, step, device, rank)
    if dist is not None:
        flag = torch.tensor([1 if ok else 0], device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        ok = bool(flag.item())
    if not ok:
        log("SELF-CHECK FAILED: " + detail)
        if dist is not None:
            dist.destroy_process_group()
        flush_c_stdout()
        if rank == 0:
            print(json.dumps({"metric": "256x256 images/sec, 8-step MaskGIT decode (vit-s-vqgan), whole job", "value": None,
                              "self_check": "FAILED: timed configuration differs from the eager single-stream loop", "n_gpus": world}),
                  flush=True)
        return 1
    log("warm-up")
    free_running = getattr(step, "joins", False) and STREAMS > 1
    for i in range(args.warmup):
        if free_running and dist is not None:
            gather_lanes(step(i, join=False))           # same code path as the timed loop (RCCL init, receive buffers)
        else:
            gather(step(i))
    drain_gathers()
    log("timed region")
    if args.dtype == "bf16":
        from paintmind_amd import ops as _ops
        _ops.attention_fallbacks(reset=True, device=device)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(device)
    power = PowerSampler(local_rank).start() if rank == 0 else None
    ru0 = resource.getrusage(resource.RUSAGE_SELF)      # process-wide: the main thread, lane threads, the HIP runtime's helpers
    th0 = thread_cpu_seconds()
    t0 = time.perf_counter()
    # The host stays at most PACE steps ahead of the GPU: before step i is enqueued it waits until step i - PACE has finished,
    # and it waits QUIETLY (event.query() + a 1 ms sleep: on this ROCm a hipEventSynchronize spins even on a blocking event
    # unless the whole device context is switched to blocking sync, which the profiler and the host-paced drop-in path both
    # dislike).  Unpaced, the launch calls of a long run block on the full hardware queue and that wait spins as well: a whole
    # core per rank for nothing.  Two steps (0.26 s of queued GPU work) of slack keep the GPU fed.
    def wait_quietly(events):
        for ev in events:
            while not ev.query():
                time.sleep(0.001)

    paced = []
    for i in range(args.steps):
        if len(paced) >= PACE:
            wait_quietly(paced.pop(0))
        if free_running:
            parts = step(args.warmup + i, join=False)   # lanes keep their own stream order; device-wide sync below joins them
            if dist is not None:
                gather_lanes(parts)
            evs = []
            for _, _, st in parts:
                ev = torch.cuda.Event()
                ev.record(st)
                evs.append(ev)
            paced.append(evs)
        else:
            gather(step(args.warmup + i))
            ev = torch.cuda.Event()
            ev.record()
            paced.append([ev])
    host_enqueue = time.perf_counter() - t0             # wall time until the last step is enqueued: INCLUDES the time the launch
    ru1 = resource.getrusage(resource.RUSAGE_SELF)      # calls block on a full hardware queue (back-pressure, not host work)
    for evs in paced:
        wait_quietly(evs)                               # the remaining <= PACE steps, without spinning
    drain_gathers()
    torch.cuda.synchronize(device)
    own_elapsed = time.perf_counter() - t0              # this rank's own K steps (before waiting for the slowest rank)
    ru2 = resource.getrusage(resource.RUSAGE_SELF)
    th1 = thread_cpu_seconds()
    power_summary = power.summary() if power is not None else None
    by_thread = sorted(((th1[t][1] - th0.get(t, (None, 0.0))[1], "main" if t == os.getpid() else th1[t][0]) for t in th1), reverse=True)
    main_cpu = th1.get(os.getpid(), (None, 0.0))[1] - th0.get(os.getpid(), (None, 0.0))[1]
    cpu_enqueue = (ru1.ru_utime + ru1.ru_stime) - (ru0.ru_utime + ru0.ru_stime)
    cpu_total = (ru2.ru_utime + ru2.ru_stime) - (ru0.ru_utime + ru0.ru_stime)   # incl. the closing synchronize (a spin or a sleep)
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    per_rank = [own_elapsed]
    per_rank_cpu = [(cpu_enqueue, cpu_total)]
    if dist is not None:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        mine = torch.tensor([own_elapsed, cpu_enqueue, cpu_total], device=device, dtype=torch.float64)
        allt = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allt, mine)
        per_rank = [float(x[0].item()) for x in allt]
        per_rank_cpu = [(float(x[1].item()), float(x[2].item())) for x in allt]
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
                   "hip_graph": bool(USE_GRAPH), "concurrent_micro_batches": STREAMS,
                   # what a rank of an N > 1 job runs (top of this file): the eager loop under AMD_DIRECT_DISPATCH=0
                   "dispatch_mode": "AMD_DIRECT_DISPATCH=" + os.environ.get("AMD_DIRECT_DISPATCH", "1 (default)"),
                   "rank_mode": bool(_RANK_MODE)},
        "images_per_s_per_gpu": round(value / world, 3),
        # workgroups of the bf16 attention kernel that left its fixed-reference fast path in the timed steps (rank 0)
        "attention_fallbacks": _ops.attention_fallbacks(reset=True, device=device) if args.dtype == "bf16" else None,
        "self_check": "ok", "self_check_detail": detail,
        "rccl_ranks": dist.get_world_size() if dist is not None else 0,
        "per_rank_images_per_s": [round(B * args.steps / e, 3) for e in per_rank],
        # HOST COST of one step, measured (getrusage for the process, /proc/self/task for the threads), over the K timed steps
        # including the closing synchronize.  host_main_thread_cpu_ms_per_step: the thread that enqueues (graph replays of the
        # lanes, parameter copies, event records, pacing) -- the path's own host work.  host_cpu_ms_per_step: the whole
        # process; the difference is the HIP runtime's helper threads, one of which busy-polls for as long as GPU work is
        # outstanding (a full core whatever the wait policy).  host_enqueue_wall_ms_per_step: wall time until the last step is
        # enqueued; it INCLUDES the pacing sleeps (the host stays <= PACE steps ahead), so it approaches ms_per_step.
        "host_main_thread_cpu_ms_per_step": round(main_cpu / args.steps * 1e3, 3),
        "host_cpu_ms_per_step": round(cpu_total / args.steps * 1e3, 3),
        "host_cpu_fraction_of_one_core": round(cpu_total / max(own_elapsed, 1e-9), 4),
        "host_enqueue_wall_ms_per_step": round(host_enqueue / args.steps * 1e3, 3),
        "host_cpu_ms_per_step_by_thread": [{"thread": nm, "cpu_ms_per_step": round(sec / args.steps * 1e3, 2)} for sec, nm in by_thread[:4] if sec > 0],
        "power": power_summary,
        # energy of one step = mean package power in the timed region x its duration: at the 1.35-1.40 kW cap the step is
        # energy-limited, and joules -- flops at ~0.9 pJ, HBM bytes at ~0.13 nJ, idle ~240 W -- is the currency
        # for deciding which lever is worth building; joules_per_image is the per-GPU figure of merit next to images/s
        "joules_per_step": round(power_summary["package_watts_mean"] * ms_per_step / 1e3, 1) if power_summary else None,
        "joules_per_image": round(power_summary["package_watts_mean"] * ms_per_step / 1e3 / B, 3) if power_summary else None,
        "per_rank_host_cpu_ms_per_step": [round(c[1] / args.steps * 1e3, 3) for c in per_rank_cpu],
        # the path's only collective: finished images of one step -> rank 0 (0 without a process group)
        "gather_bytes_per_step_per_rank": (B * 3 * 256 * 256 * 4 if cfg0 is None else B * 3 * img_px * img_px * 4) if dist is not None else 0,
    }

    log(f"timed region done: {ms_per_step:.1f} ms/step")
    pipeline = WORKLOADS[args.workload][0] is not None
    if rank == 0 and not args.no_roofline:
        # per-family kernel time of ONE more step, bracketed by hipEvents on the launch stream
        kinds = {}
        gf, af, sample_bytes = work_per_step(args.workload, decode_every_step=not args.final_decode_only, kinds=kinds)
        ops.timing_reset()
        ops.timing_enable(True)
        if pipeline:
            step(10_000, streams=1)                    # timing on => one stream, eager loop, every launch bracketed
        else:
            step(10_000)
        torch.cuda.synchronize(device)
        ops.timing_enable(False)
        fam = {f: ops.timing_get(f) for f in ("gemm", "attention", "layernorm", "sample", "vq", "rowops", "gemm_heads", "gemm_swiglu",
                                              "gemm_resid", "gemm_resid2b", "gemm_plain")}
        n_g, ms_g = fam["gemm"]
        n_a, ms_a = fam["attention"]
        n_s, ms_s = fam["sample"]
        peak = PEAK_BF16_TFLOPS if args.dtype == "bf16" else PEAK_F32_TFLOPS
        ach_g = gf / (ms_g * 1e-3) / 1e12 if ms_g > 0 else 0.0
        ach_a = af / (ms_a * 1e-3) / 1e12 if ms_a > 0 else 0.0
        # In-process CALIBRATION of this box: the library's own 8192^3 bf16 GEMM (a long-K, MFMA-bound kernel whose rate moves
        # with the box's sustained matrix clock: 1.40-1.65 PFLOP/s over the pool).  Dividing a kernel's achieved rate by
        # calibration / CALIBRATION_REFERENCE removes the box-to-box spread when two runs are compared (profiles/ vs the driver).
        cal = gemm_calibration(device) if args.dtype == "bf16" else None
        if cal:
            result["calibration"] = cal
        total_ms = sum(fam[f][1] for f in ("gemm", "attention", "layernorm", "sample", "vq", "rowops")) or 1.0

        def block(kernel, bound, flops, ms, n, traffic=None, hbm_bytes=None):
            ach = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            out = {"kernel": kernel, "bound": bound, "share_of_gpu_time": round(ms / total_ms, 4), "launches": n,
                   "avg_launch_ms": round(ms / max(n, 1), 4), "algorithmic_gflop_per_launch": round(flops / max(n, 1) / 1e9, 2)}
            if bound == "hbm" and hbm_bytes:
                gbs = hbm_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                out.update({"achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                            "algorithmic_MB_per_launch": round(hbm_bytes / max(n, 1) / 1e6, 1), "tflops": round(ach, 1)})
            else:
                out.update({"achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4)})
            if cal:
                out["frac_calibrated"] = round(out["frac"] * CALIBRATION_REFERENCE_TFLOPS / cal["gemm_8192_bf16_tflops"], 4)
            out["traffic"] = traffic
            return out

        # `roofline`: the dominant single kernel = the fused attention kernel (the largest share of GPU time of any one
        # kernel).  Algorithmic work per launch: 4 * Nq * Nkv * inner flops; the row-sum MFMAs the
        # kernel adds (1/8 more) are NOT counted.  `roofline_gemm_family`: all GEMM launches of the step (the largest share of
        # the step as a FAMILY, spread over three kernels); `roofline_by_kernel`: every kernel kind against its own bound.
        gemm_block = block("GEMM family (gemm256_kernel + gemm2b_kernel + gemm_nt_kernel, all launches of one step)", "mfma", gf, ms_g, n_g,
                           pmc_traffic(args.workload, args.dtype, "gemm"))
        attn_block = block("attention_bf16_kernel (softmax(QK^T)V, all launches of one step: stage-2 self-attention x2 per layer + ViT decoder)"
                           if args.dtype == "bf16" else "attention_kernel<float>", "mfma", af, ms_a, n_a,
                           pmc_traffic(args.workload, args.dtype, "attention"))
        if pipeline and n_a > 0:
            result["roofline"] = attn_block
            result["roofline_gemm_family"] = gemm_block
        else:
            result["roofline"] = gemm_block
        names = {"gemm_heads": ("gemm256_kernel<EPI_HEADS>: q|k|v projections, head-split epilogue, LayerNorm folded", "mfma"),
                 "gemm_swiglu": ("gemm256_kernel<EPI_SWIGLU>: FFN w12 with the SiLU gate in the epilogue, LayerNorm folded", "mfma"),
                 "gemm_resid": ("gemm256_kernel<EPI_STD, bf16 hi/lo>: FFN w3 residual producer (+ token / post-quant projections)", "mfma"),
                 "gemm_resid2b": ("gemm2b_kernel: attention out-projection residual producer (A + residual pair in + pair out)", "hbm"),
                 "gemm_plain": ("gemm256_kernel<EPI_STD, f32>: to_logits (+ the decoder's 192-wide pixel projection)", "mfma")}
        by_kernel = [dict(attn_block, traffic=None)] if n_a > 0 else []
        for f, (label, bound) in names.items():
            n_f, ms_f = fam[f]
            if n_f:
                by_kernel.append(block(label, bound if args.dtype == "bf16" else "mfma", kinds.get(f, 0), ms_f, n_f,
                                       hbm_bytes=kinds.get("bytes_" + f[5:]) if args.dtype == "bf16" else None))
        by_kernel.sort(key=lambda b: -b["share_of_gpu_time"])
        for b in by_kernel:
            b.pop("traffic", None)
        if args.workload == DEFAULT_WORKLOAD and args.dtype == "bf16" and not args.no_extra and os.environ.get("PM_BENCH_NO_POWER") != "1":
            # the power story made checkable per kernel: the clock the part grants each kernel kind, and the roofline fraction
            # against THAT clock next to the nominal one (which stays the contract)
            log("per-kernel shader clock probe")
            clocks = kernel_clock_probe(device)
            fam_of = (("gemm2b", "gemm_resid2b"), ("EPI_HEADS", "gemm_heads"), ("EPI_SWIGLU", "gemm_swiglu"), ("bf16 hi/lo", "gemm_resid"),
                      ("f32>", "gemm_plain"), ("attention_bf16_kernel", "attention"))
            for b in by_kernel:
                fam_key = next((v for k_, v in fam_of if k_ in b["kernel"]), None)
                ck = clocks.get(fam_key)
                if ck:
                    b.update(ck)
                    if b["bound"] == "mfma" and ck["sclk_mhz_alone"] > 0:
                        b["frac_at_measured_clock"] = round(b["frac"] * 2400.0 / ck["sclk_mhz_alone"], 4)
            result["roofline_by_kernel_note"] = ("sclk_mhz_alone / package_watts_alone / us_per_launch_alone: the kernel looped alone at its bench "
                                                 "shape for 1.2 s under rocm-smi (cap 1400 W, nominal 2400 MHz); frac_at_measured_clock = frac x 2400 / "
                                                 "sclk_mhz_alone; frac against the nominal peak stays the contract")
            if "error" in clocks:
                result["roofline_by_kernel_note"] += "; probe error: " + clocks["error"]
        result["roofline_by_kernel"] = by_kernel
        result["kernel_families"] = {
            f: {"launches": fam[f][0], "ms": round(fam[f][1], 3)} for f in fam}
        if ms_a > 0:
            result["kernel_families"]["attention"]["tflops"] = round(ach_a, 2)
            result["kernel_families"]["attention"]["frac_of_bf16_peak"] = round(ach_a / peak, 4)
        if ms_g > 0:
            result["kernel_families"]["gemm"]["tflops"] = round(ach_g, 2)
        if ms_s > 0 and sample_bytes:
            # sampling + re-masking launches against the bytes the sampling kernel reads (block statistics + top-k blocks)
            result["kernel_families"]["sample"]["read_GBps"] = round(sample_bytes / (ms_s * 1e-3) / 1e9, 1)
        result["end_to_end_tflops_per_gpu"] = round((gf + af) / (ms_per_step * 1e-3) / 1e12, 2)
    extra = {}
    if rank == 0 and world == 1 and pipeline and not args.final_decode_only:
        # secondary number (NOT the headline): same loop, but only the image of the last step is decoded
        alt = make_step(args.workload, model, device, rank, decode_every_step=False)
        dt = time_steps(alt, device, 2, 1, 3, STREAMS > 1)
        extra["final_decode_only_images_per_s"] = round(B / dt, 2)
    if rank == 0 and world == 1 and pipeline and not args.final_decode_only and not args.no_extra:
        log("drop-in Pipeline.generate() with default arguments")
        try:
            dg = dropin_generate(args.workload, model, device, max(2, min(args.steps, 5)))
            # save_interval=1 returns (and copies to the host) the image of EVERY step: the headline's work plus D2H
            extra["dropin_generate_images_per_s"] = dg["save_interval_1"]["images_per_s"]
            extra["dropin_generate_vs_headline"] = round(dg["save_interval_1"]["images_per_s"] / value, 4)
            extra["dropin_generate"] = dg
        except Exception as e:
            extra["dropin_generate"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and pipeline and not args.no_extra and args.workload == DEFAULT_WORKLOAD:
        try:
            extra["small_batch_latency"] = small_batch_latency(args.workload, model, device)
        except Exception as e:
            extra["small_batch_latency"] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0 and world == 1 and not args.no_extra and args.workload == DEFAULT_WORKLOAD and args.dtype == "bf16":
        del step, model
        torch.cuda.empty_cache()
        for name, dt_name, k in EXTRA_WORKLOADS:
            key = name if dt_name == "bf16" else f"{name}-{dt_name}-verify"
            log(f"extra workload {key}")
            try:
                extra[key] = extra_workload(name, dt_name, k, device)
            except Exception as e:                      # an extra never takes the headline down
                extra[key] = {"error": f"{type(e).__name__}: {e}"}
    if extra:
        result["extra"] = extra
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.workload)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdout()
    if rank == 0:
        print(json.dumps(result), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
