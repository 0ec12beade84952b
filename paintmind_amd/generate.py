"""Pipeline: frozen VQGAN + text tower + CondTransformer and the MaskGIT decode loop
(reference paintmind/generate.py:49-236), driven through the native engine.

Kept 1:1 with the reference: constructor wiring and parameter names, ``to_latent`` /
``tokens2logits`` / ``ids2tokens`` / ``sample`` / ``generate`` / ``inpaint`` / ``outpaint`` signatures,
return structures and the quirks listed in SURVEY.md section 8(a) (image decoded from the predictions at
ALL positions, confidence from the unfiltered softmax, >= 1 token re-masked on the last step,
``generate`` returning only the steps with ``step % save_interval == 0`` as CPU tensors).
The masked-token objective (forward / loss / random_masking, generate.py:78-146) is built forward-only: no backward.
"""
import math
import concurrent.futures
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops, packing
from .config import ver2cfg
from .engine import S2Engine
from .stage2 import CondTransformer


def exists(x):
    return x is not None


def _draw_seed():
    """a Philox seed from the torch CPU generator: governed by torch.manual_seed like the reference's noise"""
    return int(torch.randint(0, 2 ** 62, (1,)).item())


def mask_schedule(ratio):
    """cosine schedule, evaluated in float64 like the reference (generate.py:25-26)."""
    return np.cos(math.pi / 2. * ratio)


def num_token_masked(mask_ratio, num_tokens):
    """generate.py:175 -- max(int(mask_ratio * num_tokens), 1); accepts numpy scalars, tensors, floats."""
    r = mask_ratio * num_tokens
    r = r.item() if hasattr(r, "item") else r
    return max(int(r), 1)


class _PinnedPool:
    """Pinned host buffers for the images generate() returns.  Page-locking a fresh 400 MB allocation costs ~35 ms (more
    than the copies themselves), so buffers are kept and handed out again -- but only once nothing the caller received
    still aliases them: every returned tensor (and any view / numpy alias of it) holds a reference to the buffer's
    storage, so a use count of 2 (the pool's tensor + the probe) means the earlier results are gone.  Results a caller
    keeps stay valid for ever; their buffer simply leaves the pool."""
    MAX_KEPT = 4

    def __init__(self):
        self.bufs = []

    @staticmethod
    def _free(t):
        # private torch API; without it a buffer is never handed out twice (every call page-locks a fresh one: slower, safe)
        use_count = getattr(torch._C, "_storage_Use_Count", None)
        if use_count is None:
            return False
        try:
            return use_count(t.untyped_storage()._cdata) <= 2
        except Exception:
            return False

    def discard(self, t):
        """forget a buffer whose contents may still be written by copies in flight (a lane failed): it is never reused"""
        self.bufs = [b for b in self.bufs if b.untyped_storage().data_ptr() != t.untyped_storage().data_ptr()]

    def get(self, shape):
        n = 1
        for d in shape:
            n *= int(d)
        for t in self.bufs:
            if t.numel() == n and self._free(t):
                return t.view(shape)
        t = torch.empty(n, dtype=torch.float32, pin_memory=True)
        self.bufs = [b for b in self.bufs if not (self._free(b) and b.numel() != n)][-(self.MAX_KEPT - 1):] + [t]
        return t.view(shape)


_pinned_pool = _PinnedPool()
_lane_threads = None


def _lane_thread_pool(n):
    """process-wide worker threads that drive concurrent micro-batch lanes (kept out of the Module: it stays copyable)"""
    global _lane_threads
    if _lane_threads is None or _lane_threads._max_workers < n:
        from concurrent.futures import ThreadPoolExecutor
        _lane_threads = ThreadPoolExecutor(max_workers=max(n, 2), thread_name_prefix="pm-lane")
    return _lane_threads


T5_VERSION = {'t5-l': 'google/flan-t5-large', 't5-xl': 'google/flan-t5-xl', 't5-xxl': 'google/flan-t5-xxl'}
T5_TXT_DIM = {'t5-l': 1024, 't5-xl': 2048}      # no 't5-xxl' entry, as in the reference (generate.py:53)


# generate(): smallest B * tokens * width that runs as two concurrent micro-batch lanes by default (bf16).  Below it ONE lane, whose
# per-step decode is deferred beside the next step's tower, is faster; measured crossovers (tools/small_batch_lanes_ab.py,
# profiles/r05_g_*): d512 between 32 and 36 images, d768 between 20 and 22, d1024 between 16 and 20 -- 16-19 M in this unit.
LANES_MIN_WORK = 17_000_000


class Pipeline(nn.Module):
    def __init__(self, config, stage1_pretrained=True, stage1_checkpoint_path=None, text_model=None):
        super().__init__()
        from .factory import create_model
        self.vqgan = create_model(arch='vqgan', version=config.stage1, pretrained=stage1_pretrained,
                                  checkpoint_path=stage1_checkpoint_path)
        self.vqgan.freeze()

        context_dim = getattr(config, "context_dim", None) or T5_TXT_DIM[config.t5]
        if text_model is not None:
            self.text_model = text_model
        elif getattr(config, "text_model", "t5") == "none":
            from .modules.encoder import SyntheticTextEmbedder
            self.text_model = SyntheticTextEmbedder(context_dim)
        else:
            from .modules.encoder import T5TextEmbedder
            self.text_model = T5TextEmbedder(version=T5_VERSION[config.t5], freeze=True)

        vq_cfg = ver2cfg[config.stage1]
        self.image_size = vq_cfg['enc']['image_size']
        self.patch_size = vq_cfg['enc']['patch_size']
        self.num_tokens = (self.image_size // self.patch_size) ** 2
        self._width = config.dim

        self.transformer = CondTransformer(
            vq_cfg['embed_dim'], config.dim, self.num_tokens, config.dim_head, config.mlp_dim,
            config.num_head, config.depth, config.dropout, context_dim, vq_cfg['n_embed'],
        )
        self.mask_token = nn.Parameter(torch.zeros(1, vq_cfg['embed_dim']))
        self.mask_token_id = vq_cfg['n_embed']
        nn.init.normal_(self.mask_token, std=.02)
        self._pm_dtype = torch.float32
        self._engine = None

    @property
    def image_shape(self):
        """(C, H, W) of the images this pipeline returns (used by dist.generate_sharded for ranks with an empty shard)"""
        return (self.vqgan.encoder.to_patch_embedding[0].in_channels, self.image_size, self.image_size)

    # -- precision / engines ------------------------------------------------------------------------
    def set_compute_dtype(self, dtype):
        self.vqgan.set_compute_dtype(dtype)
        for m in self.transformer.modules():
            m._pm_dtype = dtype
        self._pm_dtype = dtype
        return self

    @property
    def compute_dtype(self):
        if torch.is_autocast_enabled('cuda') and torch.get_autocast_dtype('cuda') == torch.bfloat16:
            return torch.bfloat16
        return self._pm_dtype

    def engine(self):
        """native stage-2 handle for the current compute dtype; one cached per dtype, rebuilt when a parameter's
        (data_ptr, version) changes.  After edits through `p.data` call invalidate_engines()."""
        dtype = self.compute_dtype
        cb = self.vqgan.quantize.embedding.weight
        stamp = (packing.params_fingerprint(self.transformer), cb.data_ptr(), cb._version, self.mask_token.data_ptr(),
                 self.mask_token._version)
        cache = self._engine if isinstance(self._engine, dict) else {}
        hit = cache.get(dtype)
        if hit is None or hit[0] != stamp:
            cache[dtype] = hit = (stamp, S2Engine(self.transformer, cb, self.mask_token, dtype))
            self._engine = cache
        return hit[1]

    def invalidate_engines(self):
        """drop every packed weight copy / captured graph of this pipeline and its VQGAN"""
        self._engine = None
        self._lane_cache = None
        self._copy_streams = None
        self.vqgan.invalidate_engines()

    def from_pretrained(self, path):
        return self.load_state_dict(torch.load(path, map_location="cpu"))

    # -- masked-token objective, forward only (generate.py:78-146) ---------------------------------
    # The HIP path has no backward: these return tensors without a grad_fn, for validation loss and
    # for checking a training run's forward numerics.  Optimisation stays with the reference trainer.
    @torch.no_grad()
    def random_masking(self, x, mask_ratio, noise=None):
        """(x [B,L,D], ratio) -> (x with masked positions replaced by mask_token, mask [B,L], 1 = masked)
        (generate.py:78-110).  `noise` [B,L]: the uniforms the reference draws at :89 (default: torch.rand)."""
        B, L, _ = x.shape
        len_keep = L - max(int(L * mask_ratio), 1)
        if noise is None:
            noise = torch.rand(B, L, device=x.device)
        return ops.random_mask(x.float().contiguous(), noise.float().contiguous(),
                               self.mask_token.data.float().reshape(-1).contiguous(), len_keep)

    @torch.no_grad()
    def loss(self, logit, label, masks):
        """label-smoothed (0.1) cross entropy averaged over the masked positions (generate.py:112-125)"""
        V = logit.shape[-1]
        out, _ = ops.masked_ce(logit.float().reshape(-1, V).contiguous(), label.reshape(-1).contiguous(),
                               masks.float().reshape(-1).contiguous(), 0.1)
        return out.reshape(())

    @torch.no_grad()
    def forward(self, img, text=None, mask_ratio=0.75, noise=None):
        """generate.py:136-146: encode -> random masking -> stage-2 logits -> masked cross entropy"""
        x, ids, text = self.to_latent(img, text)
        x, mask = self.random_masking(x, mask_ratio, noise)
        return self.loss(self.tokens2logits(x, text), ids, mask)

    # -- inference API ------------------------------------------------------------------------------
    @torch.no_grad()
    def to_latent(self, img, text=None):
        x, _, indices = self.vqgan.encode(img)
        if exists(text):
            text = self.text_model(text)
        return x, indices, text

    def _on_cpu(self):
        return self.mask_token.device.type == "cpu"

    def tokens2logits(self, token, text=None):
        if self._on_cpu():
            return self.transformer(token, text)             # plain-torch operators of this package (no HIP engine on the CPU)
        return self.engine().forward(token, text)

    @torch.no_grad()
    def ids2tokens(self, ids):
        """lookup in cat(RAW codebook, mask_token) (generate.py:148-157)."""
        table = torch.cat((self.vqgan.quantize.embedding.weight.data.float(), self.mask_token.data.float())).contiguous()
        if self._on_cpu():
            return table[ids]
        rows = ops.embed_rows(table, ids.contiguous().reshape(-1), table.shape[1], torch.float32)
        return rows.reshape(ids.shape + (table.shape[1],))

    @torch.no_grad()
    def sample(self, ids, mask_ratio, text=None, topk=1, temperature=1, noise=None, seed=None, step=0, image_base=0,
               guidance_scale=None):
        """One MaskGIT step (generate.py:159-181) -> (ids', img).

        ``noise``: optional uniform(0,1) tensor shaped like the logits (B,N,V) -- the parity hook for the
        reference's ``torch.zeros_like(t).uniform_(0,1)``; without it a counter-based Philox stream keyed
        by (seed, step, image_base + image index, position, class) is used.  ``seed=None`` draws a fresh
        seed from the torch generator on every call, like the reference's fresh global-RNG noise
        (generate.py:40-46), so a caller looping over sample() never reuses uniforms.

        ``guidance_scale`` (extension; None = the reference's behaviour): the step's logits become
        ``uncond + guidance_scale * (cond - uncond)`` with ``uncond = transformer(tokens, None)``, the path the reference
        trains by dropping the text 10 % of the time (utils/trainer.py:379,387-388) but never uses when sampling.  Both
        forwards share the step's token lookup; everything after the logits is the reference's step unchanged.
        """
        nm = num_token_masked(mask_ratio, self.num_tokens)
        if guidance_scale is not None and text is None:
            raise ValueError("guidance_scale needs a text condition (text=None IS the unconditional branch)")
        if self._on_cpu():
            return self._sample_cpu(ids, nm, text, topk, temperature, noise, seed, guidance_scale)
        if seed is None:
            seed = _draw_seed()
        eng = self.engine()
        ids = ids.to(eng.device, torch.int64).clone().contiguous()
        ids, img, _, _ = eng.sample(self.vqgan.engine(), ids, text, topk, temperature, nm, noise=noise, seed=seed, step=step,
                                    image_base=image_base, want_img=True, guidance_scale=guidance_scale)
        return ids, img

    def _sample_guided_composed(self, ids, nm, text, topk, temperature, noise, seed, step, image_base, scale):
        """the guided step composed from the operator-level C ABI (two pmhip_s2_forward + pmhip_guidance_combine +
        pmhip_sample_rows + decode + pmhip_remask): the same kernels, in the same order, as pmhip_pipeline_sample_guided.
        Not on any product path: the bit-identity reference of tests/test_gpu_model.py for the native guided step / loop."""
        eng = self.engine()
        ids = ids.to(eng.device, torch.int64).clone().contiguous()
        B, N = ids.shape
        tok = self.ids2tokens(ids)
        cond = eng.forward(tok, text)
        uncond = eng.forward(tok, None)
        logits = ops.guidance_combine(cond, uncond, scale, out=cond)
        if noise is not None:
            noise = noise.to(eng.device, torch.float32).reshape(B * N, -1).contiguous()
        pred, merged, score = ops.sample_rows(logits.reshape(B * N, -1), ids.reshape(-1), self.mask_token_id, topk, temperature,
                                              noise=noise, seed=seed, step=step, row_base=image_base * N)
        img = self.vqgan.decode_from_indice(pred.reshape(B, N))   # decoded from the predictions at ALL positions (:165)
        ids = ops.remask(merged.reshape(B, N), score.reshape(B, N), nm, self.mask_token_id)
        return ids, img

    def _sample_cpu(self, ids, nm, text, topk, temperature, noise, seed, guidance_scale=None):
        """generate.py:159-181 in plain torch for a pipeline that lives on the CPU.  The noise is drawn from the torch CPU
        generator like the reference's (`seed` re-seeds a private generator; `noise` overrides it); ties in top-k / argmax
        follow torch."""
        tok = self.ids2tokens(ids)
        logits = self.tokens2logits(tok, text)
        if guidance_scale is not None:
            uncond = self.tokens2logits(tok, None)
            logits = torch.addcmul(uncond, logits - uncond, torch.tensor(float(guidance_scale)))
        val, ind = logits.topk(topk, dim=-1)
        filtered = torch.full_like(logits, float("-inf")).scatter_(2, ind, val)
        if noise is None:
            g = None if seed is None else torch.Generator().manual_seed(int(seed) & (2 ** 63 - 1))
            noise = torch.rand(logits.shape, generator=g)
        gumbel = -torch.log((-torch.log(noise.clamp(min=1e-20))).clamp(min=1e-20))
        pred = (filtered / max(temperature, 1e-10) + gumbel).argmax(dim=-1)
        img = self.vqgan.decode_from_indice(pred)            # decoded from the predictions at ALL positions (:165)
        is_mask = ids == self.mask_token_id
        ids = torch.where(is_mask, pred, ids)
        scores = 1 - logits.softmax(dim=-1).gather(2, pred[..., None])[..., 0]
        scores = scores.masked_fill(~is_mask, -1e5)
        ids = ids.scatter(1, scores.topk(nm, dim=-1).indices, self.mask_token_id)
        return ids, img

    def _generate_cpu(self, text, context, timesteps, temperature, topk, save_interval, seed, return_ids):
        B = len(text)
        ids = torch.full((B, self.num_tokens), self.mask_token_id, dtype=torch.long)
        imgs = []
        for step in range(timesteps):
            masked_r = mask_schedule((step + 1) / timesteps)
            ids, img = self._sample_cpu(ids, num_token_masked(masked_r, self.num_tokens), context, topk,
                                        temperature * (1 - step / timesteps), None, None if seed is None else seed + step)
            if step % save_interval == 0:
                imgs.append(img)
        return (imgs, ids) if return_ids else imgs

    def _schedule(self, timesteps, temperature):
        temps, nmask = [], []
        for step in range(timesteps):
            progress = (step + 1) / timesteps
            masked_r = mask_schedule(progress)
            temps.append(temperature * (1 - step / timesteps))
            nmask.append(num_token_masked(masked_r, self.num_tokens))
        return temps, nmask

    @torch.no_grad()
    def _lanes(self, k):
        """k (engine pair, stream) lanes for concurrent micro-batches; lane 0 is the primary engines"""
        eng, vq = self.engine(), self.vqgan.engine()
        cache = getattr(self, "_lane_cache", None)
        if cache is None or cache[0] is not eng or cache[1] is not vq:
            cache = (eng, vq, [(eng, vq, self._new_lane_stream(eng.device, 0, k))])
            self._lane_cache = cache
        lanes = cache[2]
        while len(lanes) < k:
            lanes.append((eng.clone(), vq.clone(), self._new_lane_stream(eng.device, len(lanes), k)))
        return lanes[:k]

    @staticmethod
    def _new_lane_stream(device, lane, n_lanes):
        """the HIP stream of one lane (a hook: tools/cu_mask_lanes.py replaces it with CU-masked streams for an experiment)"""
        return torch.cuda.Stream(device=device)

    def generate_ids(self, context, B, timesteps, temperature, topk, decode_flags, seed, image_base=0, use_graph=False, streams=1,
                     join=True, wait_current=True, host=None, guidance_scale=None, ids0=None):
        """The decode loop on device tensors: returns (ids [B,N], imgs [n_decoded,B,C,H,W] or None).

        streams > 1 (or a tuple of micro-batch sizes): the batch is cut into contiguous micro-batches that run CONCURRENTLY on separate HIP
        streams (own native handle + workspace each, same weights): memory-bound kernels of one micro-batch overlap
        the MFMA-bound kernels of another.  The sampling RNG is keyed by the global image index, so the result is
        identical to streams=1.
        join=False returns the per-lane results [(ids, imgs, stream), ...] without making the current stream wait (the
        caller joins); wait_current=False does not make the lanes wait for work already queued on the current stream
        (only valid when `context` is None or was produced before the lanes last synchronised with it).
        host = (pinned [n_decoded, B, C, H, W] float32 tensor, [copy stream per lane]): the decoded images go straight to
        the host buffer (every lane fills its rows, each image as soon as it is complete, on its own copy stream); no
        device image tensor is returned.
        ids0 (streams = 1 only): start from these ids [B, N] int64 instead of the all-mask state (the region loops of inpaint / outpaint)."""
        eng = self.engine()
        temps, nmask = self._schedule(timesteps, temperature)
        if isinstance(streams, (list, tuple)):           # explicit micro-batch sizes, e.g. (32, 16, 16)
            sizes = [int(x) for x in streams]
            if sum(sizes) != B or min(sizes) < 1:
                raise ValueError(f"micro-batch sizes {sizes} must be positive and sum to the batch size {B}")
            bounds = [(sum(sizes[:i]), sum(sizes[:i + 1])) for i in range(len(sizes))]
            streams = len(sizes)
        else:
            streams = max(1, min(int(streams), B))
            bounds = None
            if streams == 2 and B >= 8:
                # two lanes of EQUAL size run the same kernel sequence in lockstep and meet in the same (MFMA- or HBM-bound)
                # kernel all the time; one image of difference lets them drift apart (round 3: 445-447 -> 450-453 images/s).
                # Re-measured with the round-5 kernels (profiles/r05_g_*): equal lanes are within +-1 % of this end to end on
                # every workload and box tried (and the per-launch attention time in the bench's brackets is the same): kept.
                bounds = [(0, B // 2 + 1), (B // 2 + 1, B)]
        if ids0 is not None and streams != 1:
            raise ValueError("generate_ids: ids0 needs streams=1")
        if streams == 1:
            ids = torch.full((B, self.num_tokens), self.mask_token_id, dtype=torch.long, device=eng.device) if ids0 is None \
                else ids0.to(eng.device, torch.long).clone()
            return eng.generate(self.vqgan.engine(), ids, context, temps, nmask, decode_flags, topk, seed=seed,
                                image_base=image_base, use_graph=use_graph,
                                host=None if host is None else (host[0], 0, host[1][0]), want_device_imgs=host is None,
                                guidance_scale=guidance_scale)
        from .dist import shard_range
        cur = torch.cuda.current_stream(eng.device)
        if wait_current:
            ready = torch.cuda.Event()
            ready.record(cur)

        def run_lane(i, e, v, st):
            lo, hi = bounds[i] if bounds is not None else shard_range(B, i, streams)
            if wait_current:
                st.wait_event(ready)
            with torch.cuda.device(eng.device), torch.cuda.stream(st):
                ids = torch.full((hi - lo, self.num_tokens), self.mask_token_id, dtype=torch.long, device=eng.device)
                c = None if context is None else context[lo:hi].contiguous()
                ids, imgs = e.generate(v, ids, c, temps, nmask, decode_flags, topk, seed=seed, image_base=image_base + lo,
                                       use_graph=use_graph, host=None if host is None else (host[0], lo, host[1][i]),
                                       want_device_imgs=host is None, guidance_scale=guidance_scale, concurrent_lanes=True)
            return ids, imgs, st

        lanes = self._lanes(streams)
        if host is None:
            parts = [run_lane(i, e, v, st) for i, (e, v, st) in enumerate(lanes)]
        else:
            # with a host destination the native call paces its lane from the host (it blocks between segments, see
            # pmhip_pipeline_generate), so every lane is driven by its own thread; ctypes drops the GIL during the call
            pool = _lane_thread_pool(streams)
            futs = [pool.submit(run_lane, i, e, v, st) for i, (e, v, st) in enumerate(lanes)]
            # wait for EVERY lane before a failure is reported: the other lanes keep copying into the caller's host buffer
            # until their native call returns.  That includes an exception delivered to THIS thread while it waits (a
            # KeyboardInterrupt): the lane threads are inside the native call and cannot be cancelled, so the caller's
            # cleanup (which retires the host buffer) may only run once they have all returned.
            try:
                done = [(f.exception(), f) for f in futs]
            finally:
                concurrent.futures.wait(futs)
            for err, _ in done:
                if err is not None:
                    raise err
            parts = [f.result() for _, f in done]
        if not join:
            return parts
        return self.join_lanes(parts)

    def join_lanes(self, parts):
        """make the current stream wait for the lanes and concatenate their results"""
        cur = torch.cuda.current_stream(parts[0][0].device)
        for ids, imgs, st in parts:
            cur.wait_stream(st)
            ids.record_stream(cur)                    # allocated on the lane's stream, consumed on the current one
            if imgs is not None:
                imgs.record_stream(cur)
        ids = torch.cat([p[0] for p in parts], dim=0)
        imgs = None if parts[0][1] is None else torch.cat([p[1] for p in parts], dim=1)
        return ids, imgs

    def generate(self, text, timesteps=18, temperature=1.0, topk=5, save_interval=2, seed=None, image_base=0,
                 return_ids=False, keep_on_device=False, use_graph=None, streams=None, guidance_scale=None):
        """Full decode loop (generate.py:183-198): list of (B,3,H,W) CPU tensors for steps % save_interval == 0.

        The call is the fast path by default: the loop replays captured hipGraphs (first call eager, second call captures),
        in bf16 mode a batch of at least LANES_MIN_WORK (B * tokens * width: 33 images for d512, 22 for d768, 17 for d1024) runs
        as two concurrent micro-batch lanes (a smaller one as one lane whose per-step decode overlaps the next step), and every saved image starts its copy into a
        pinned host buffer on a copy stream as soon as its step is done (the reference's blocking `img.cpu()` per saved
        step, generate.py:195-196), so only the last image's copy is exposed.  use_graph / streams override the defaults;
        results are bit-identical for every setting (tests/test_gpu_model.py).

        Resource profile of the defaults (differs from a plain eager loop): the second lane owns a clone of both native
        handles -- its own workspace (sized for its share of the batch, so the total stays about the single-stream
        workspace: 6 GiB at B = 64 with vit-s + 12L/d512), its own captured graphs and its own handle-owned image buffer;
        the lanes are driven by a process-wide thread pool (one thread per lane, blocked in the native call between
        segments).  Opt out per call with streams=1 / use_graph=False, or process-wide with the environment variables
        PMHIP_GENERATE_STREAMS=1 and PMHIP_GENERATE_GRAPH=0 (read at every call).

        The returned tensors are views of ONE pinned host buffer [n_saved, B, C, H, W] that the package reuses once no
        tensor (or numpy alias) of an earlier call is alive: keeping one image keeps the whole buffer page-locked, and an
        asynchronous `.to('cuda', non_blocking=True)` of a returned image must be synchronised before the LAST reference to
        the list is dropped.  `.clone()` a result to own ordinary pageable memory, as the reference's `img.cpu()` returns."""
        B = len(text)
        context = self.text_model(text)
        if guidance_scale is not None and context is None:
            raise ValueError("guidance_scale needs a text condition (text=None IS the unconditional branch)")
        if self._on_cpu():
            if guidance_scale is not None:
                return self._generate_guided_cpu(context, B, timesteps, temperature, topk, save_interval, seed, return_ids, guidance_scale)
            return self._generate_cpu(text, context, timesteps, temperature, topk, save_interval, seed, return_ids)
        eng = self.engine()
        if seed is None:
            seed = _draw_seed()
        if use_graph is None:
            use_graph = os.environ.get("PMHIP_GENERATE_GRAPH", "1") != "0"
        if streams is None:
            # two lanes once one lane alone fills the chip (LANES_MIN_WORK above): at B = 8..20 ONE lane -- whose decode then overlaps
            # the next step's tower -- is 4-15 % faster than two; two lanes win from 33 (d512) / 22 (d768) / 17 (d1024) images
            streams = 2 if (self.compute_dtype == torch.bfloat16 and B * self.num_tokens * self._width >= LANES_MIN_WORK) else 1
            env_streams = os.environ.get("PMHIP_GENERATE_STREAMS")
            if env_streams:
                try:
                    streams = max(1, min(int(env_streams), B))   # never more lanes than images
                except ValueError:
                    pass                                          # not a number: keep the default
        flags = [step % save_interval == 0 for step in range(timesteps)]
        if context is not None:
            context = context.to(eng.device)
        n_dec = sum(flags)
        if keep_on_device or n_dec == 0:
            ids, imgs = self.generate_ids(context, B, timesteps, temperature, topk, flags, seed, image_base=image_base,
                                          use_graph=use_graph, streams=streams, guidance_scale=guidance_scale)
            out = [] if imgs is None else list(imgs)
            return (out, ids) if return_ids else out
        vq = self.vqgan.engine()
        host = _pinned_pool.get((n_dec, B, vq.channels, vq.image_size, vq.image_size))

        n_lanes = len(streams) if isinstance(streams, (list, tuple)) else max(1, min(int(streams), B))
        cs = getattr(self, "_copy_streams", None)
        if cs is None or cs[0].device != eng.device:
            cs = self._copy_streams = []
        while len(cs) < n_lanes:
            cs.append(torch.cuda.Stream(device=eng.device))
        try:
            ids, _ = self.generate_ids(context, B, timesteps, temperature, topk, flags, seed, image_base=image_base,
                                       use_graph=use_graph, streams=streams, host=(host, cs), guidance_scale=guidance_scale)
        except BaseException:
            # a lane failed: whatever the other lanes queued may still be writing into `host`; drain it, and never hand
            # this buffer out again
            _pinned_pool.discard(host)
            try:
                torch.cuda.synchronize(eng.device)
            except Exception:
                pass
            raise
        finally:
            for c in cs[:n_lanes]:
                try:
                    c.synchronize()                          # every image has landed in the host buffer
                except Exception:
                    pass
        out = list(host)                                     # views of one pinned buffer; it is reused once all of them are gone
        return (out, ids) if return_ids else out

    @torch.no_grad()
    def _generate_guided_cpu(self, context, B, timesteps, temperature, topk, save_interval, seed, return_ids, scale):
        """generate.py:183-198 with guided steps (see `sample`) for a pipeline that lives on the CPU; same return structure.
        (On the GPU the guided loop is the native one: pmhip_pipeline_generate_guided, graph-captured and lane-able.)"""
        ids = torch.full((B, self.num_tokens), self.mask_token_id, dtype=torch.long)
        imgs = []
        for step in range(timesteps):
            nm = num_token_masked(mask_schedule((step + 1) / timesteps), self.num_tokens)
            ids, img = self._sample_cpu(ids, nm, context, topk, temperature * (1 - step / timesteps), None,
                                        None if seed is None else seed + step, scale)
            if step % save_interval == 0:
                imgs.append(img)
        return (imgs, ids) if return_ids else imgs

    def _region_loop(self, img, coord, text, timesteps, topk, temperature, keep_inside, seed=None, return_ids=False):
        if seed is None:
            seed = _draw_seed()                 # one stream per call; the step index separates the steps
        z, ids, text = self.to_latent(img, text)
        s = self.patch_size
        x, y, h, w = coord[0] // s, coord[1] // s, coord[2] // s, coord[3] // s
        g = self.image_size // s
        keep = torch.zeros(g, g, dtype=torch.bool, device=ids.device) if keep_inside else \
            torch.ones(g, g, dtype=torch.bool, device=ids.device)
        keep[y:y + h, x:x + w] = keep_inside
        keep = keep.reshape(1, -1)
        # the reference builds this with float arithmetic (ids*mask + id*(1-mask), generate.py:210,229),
        # which yields a float tensor that nn.Embedding rejects; the intended integer result is used here
        ids = torch.where(keep, ids, torch.full_like(ids, self.mask_token_id))
        if timesteps >= 2 and not self._on_cpu():
            # more than the reference's default single step: the native decode loop (graph replay by default, like generate()) from
            # these start ids, decoding only the last step -- bit-identical to the per-step composition below (tests/test_gpu_model.py)
            use_graph = os.environ.get("PMHIP_GENERATE_GRAPH", "1") != "0"
            ids, imgs = self.generate_ids(text, ids.shape[0], timesteps, temperature, topk, [False] * (timesteps - 1) + [True], seed,
                                          use_graph=use_graph, streams=1, ids0=ids)
            return (imgs[0], ids) if return_ids else imgs[0]
        return self._region_steps(ids, text, timesteps, topk, temperature, seed, return_ids)

    def _region_steps(self, ids, text, timesteps, topk, temperature, seed, return_ids=False):
        """the region loop as the reference writes it: one sample() per step (generate.py:211-216,230-235)"""
        out = None
        for step in range(timesteps):
            progress = (step + 1) / timesteps
            masked_r = mask_schedule(progress)
            cur_temp = temperature * (1 - step / timesteps)
            ids, out = self.sample(ids, mask_ratio=masked_r, text=text, topk=topk, temperature=cur_temp, seed=seed, step=step)
        return (out, ids) if return_ids else out

    @torch.no_grad()
    def inpaint(self, img, coord, text=None, timesteps=1, topk=1, temperature=0, seed=None, return_ids=False):
        """re-generate the rectangle coord=(x,y,h,w) in pixels (generate.py:200-217)."""
        return self._region_loop(img, coord, text, timesteps, topk, temperature, keep_inside=False, seed=seed,
                                 return_ids=return_ids)

    @torch.no_grad()
    def outpaint(self, img, coord, text=None, timesteps=1, topk=1, temperature=0, seed=None, return_ids=False):
        """keep the rectangle, re-generate everything else (generate.py:219-236)."""
        return self._region_loop(img, coord, text, timesteps, topk, temperature, keep_inside=True, seed=seed,
                                 return_ids=return_ids)
