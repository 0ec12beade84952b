"""Host side of the model-level C ABI: pack a module tree's weights, build the pointer tables
(pmhip_vqgan_weights / pmhip_s2_weights) and own the native handles.

Engines are cached per (parameter fingerprint, compute dtype, device): loading a checkpoint, moving
the module or editing a weight in place transparently rebuilds the packed copy.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib, ops, packing
from ._lib import LayerWeights, S2Cfg, S2Weights, TowerCfg, VqganCfg, VqganWeights, check
from .ops import _p, pm_dtype, round_up, stream_ptr


def _f32(t):
    return t.detach().float().contiguous()


class _Keep:
    """keeps packed tensors alive for as long as the native handle points at them"""

    def __init__(self):
        self.tensors = []

    def __call__(self, t):
        self.tensors.append(t)
        return C.c_void_p(t.data_ptr())


def _pack_layer(layer, dtype, keep, stage2):
    lw = LayerWeights()
    a1 = layer.attn1
    lw.ln1_g, lw.ln1_b = keep(_f32(layer.norm1.weight)), keep(_f32(layer.norm1.bias))
    lw.wqkv = keep(packing.pack_qkv(a1.to_q, a1.to_k, a1.to_v, dtype))
    lw.wo, lw.bo = keep(packing.cast(a1.to_out[0].weight, dtype)), keep(_f32(a1.to_out[0].bias))
    if stage2:
        a2 = layer.attn2
        lw.lnx_g, lw.lnx_b = keep(_f32(layer.norm2.weight)), keep(_f32(layer.norm2.bias))
        lw.wqkv2 = keep(packing.pack_qkv(a2.to_q, a2.to_k, a2.to_v, dtype))
        lw.wo2, lw.bo2 = keep(packing.cast(a2.to_out[0].weight, dtype)), keep(_f32(a2.to_out[0].bias))
        ffn_norm = layer.norm3
    else:
        ffn_norm = layer.norm2
    lw.ln2_g, lw.ln2_b = keep(_f32(ffn_norm.weight)), keep(_f32(ffn_norm.bias))
    w12p, b12p, hp = packing.pack_w12(layer.ffnet.w12, dtype)
    lw.w12p, lw.b12p = keep(w12p), keep(b12p)
    lw.w3p, lw.b3 = keep(packing.pack_w3(layer.ffnet.w3, hp, dtype)), keep(_f32(layer.ffnet.w3.bias))
    # what each residual producer adds to every row's mean (centred hi plane, include/pmhip.h)
    lw.bo_mean = float(a1.to_out[0].bias.detach().float().mean())
    lw.b3_mean = float(layer.ffnet.w3.bias.detach().float().mean())
    if stage2:
        lw.bo2_mean = float(layer.attn2.to_out[0].bias.detach().float().mean())
    if dtype == torch.bfloat16:
        # LayerNorm fold (include/pmhip.h, pmhip_lnfold): gamma-scaled copies of the weights that consume a LayerNorm
        def fold(w32, norm):
            wg, c, d = packing.ln_fold(w32, norm.weight, norm.bias, dtype)
            return keep(wg), keep(c), keep(d)
        lw.wqkv_f, lw.qkv_c, lw.qkv_d = fold(packing.pack_qkv(a1.to_q, a1.to_k, a1.to_v, torch.float32), layer.norm1)
        if stage2:
            a2 = layer.attn2
            if a2.to_k.in_features == a2.to_q.in_features:      # always true in the reference (context is pre-projected)
                lw.wqkv2_f, lw.qkv2_c, lw.qkv2_d = fold(packing.pack_qkv(a2.to_q, a2.to_k, a2.to_v, torch.float32), layer.norm2)
        lw.w12p_f, lw.w12_c, lw.w12_d = fold(packing.pack_w12(layer.ffnet.w12, torch.float32)[0], ffn_norm)
    return lw, hp


def _pack_tower(layers, dtype, keep, stage2):
    arr = (LayerWeights * len(layers))()
    hp = 0
    for i, layer in enumerate(layers):
        arr[i], hp = _pack_layer(layer, dtype, keep, stage2)
    return arr, hp


def _switch_dict(bits):
    return {"ln_fold": bool(bits & 1), "hilo": bool(bits & 2), "ln_stats": bool(bits & 4), "hilo_center": bool(bits & 8),
            "blocking_wait": bool(bits & 16), "graph_replay_off": bool(bits & 32)}


class VqganEngine:
    """native VQModel handle (pmhip_vqgan) for one (weights, dtype, device)."""

    @property
    def switches(self):
        """the PMHIP_* switches this handle latched when it was created (they are not re-read: include/pmhip.h)"""
        return _switch_dict(self.lib.pmhip_vqgan_switches(self.handle))

    def __init__(self, model, dtype):
        self.lib = _lib.load()
        dev = next(model.parameters()).device
        if dev.type != "cuda":
            raise _lib.PmhipError("paintmind_amd needs the model on a ROCm device (model.to('cuda')); no CPU fallback")
        self.device, self.dtype = dev, dtype
        keep = self.keep = _Keep()
        enc, dec, vq = model.encoder, model.decoder, model.quantize
        conv = enc.to_patch_embedding[0]
        P = enc.patch_size
        self.channels = conv.in_channels
        self.image_size = enc.image_size
        self.tokens = (enc.image_size // P) ** 2
        self.embed_dim = vq.e_dim
        with torch.cuda.device(dev):
            enc_layers, enc_hp = _pack_tower(list(enc.transformer.layers), dtype, keep, False)
            dec_layers, dec_hp = _pack_tower(list(dec.transformer.layers), dtype, keep, False)
            self._layer_arrays = (enc_layers, dec_layers)
            en, sq = vq.prepared()
            w = VqganWeights()
            w.patch_w = keep(packing.cast(conv.weight.reshape(conv.out_channels, -1), dtype))
            w.enc_pos = keep(_f32(enc.position_embedding[0]))
            w.pre_g, w.pre_b = keep(_f32(enc.norm_pre.weight)), keep(_f32(enc.norm_pre.bias))
            w.enc_layers = enc_layers
            w.prevq_w, w.prevq_b = keep(packing.cast(model.prev_quant.weight, dtype)), keep(_f32(model.prev_quant.bias))
            w.codebook_n, w.codebook_sq = keep(en), keep(sq)
            w.postq_w, w.postq_b = keep(packing.pad_cols(model.post_quant.weight, 64, dtype)), keep(_f32(model.post_quant.bias))
            w.dec_pos = keep(_f32(dec.position_embedding[0]))
            w.dec_layers = dec_layers
            w.dn_g, w.dn_b = keep(_f32(dec.norm.weight)), keep(_f32(dec.norm.bias))
            w.proj_w, w.proj_b = keep(packing.cast(dec.proj.weight, dtype)), keep(_f32(dec.proj.bias))
            torch.cuda.synchronize(dev)
        cfg = VqganCfg()
        cfg.image_size, cfg.patch_size, cfg.channels = enc.image_size, P, conv.in_channels
        cfg.n_embed, cfg.embed_dim, cfg.beta = vq.n_e, vq.e_dim, float(vq.beta)
        a_enc = enc.transformer.layers[0].attn1
        a_dec = dec.transformer.layers[0].attn1
        cfg.enc = TowerCfg(conv.out_channels, len(enc.transformer.layers), a_enc.heads, enc_hp, a_enc.dim_head)
        cfg.dec = TowerCfg(dec.proj.in_features, len(dec.transformer.layers), a_dec.heads, dec_hp, a_dec.dim_head)
        self.cfg, self.weights = cfg, w
        self.handle = C.c_void_p()
        check(self.lib.pmhip_vqgan_create(C.byref(self.handle), dev.index or 0, pm_dtype(dtype), C.byref(cfg), C.byref(w)),
              "pmhip_vqgan_create")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.pmhip_vqgan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def clone(self):
        """a second native handle (own workspace, own graphs) over the SAME packed weights: one per concurrent stream"""
        other = object.__new__(VqganEngine)
        other.__dict__.update({k: v for k, v in self.__dict__.items() if k != "handle"})
        other.handle = C.c_void_p()
        check(self.lib.pmhip_vqgan_create(C.byref(other.handle), self.device.index or 0, pm_dtype(self.dtype), C.byref(self.cfg),
                                          C.byref(self.weights)), "pmhip_vqgan_create")
        return other

    # -- entry points -------------------------------------------------------------------------------
    def _img(self, img):
        if not img.is_cuda:
            raise _lib.PmhipError("input image is on the CPU; paintmind_amd has no CPU fallback")
        return img.to(self.device, torch.float32).contiguous()

    def encode(self, img):
        img = self._img(img)
        B = img.shape[0]
        z = torch.empty(B, self.tokens, self.embed_dim, device=self.device, dtype=torch.float32)
        idx = torch.empty(B, self.tokens, device=self.device, dtype=torch.int64)
        loss = torch.empty(1, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            check(self.lib.pmhip_vqgan_encode(self.handle, _p(img), B, _p(z), _p(idx), _p(loss), stream_ptr(self.device)),
                  "pmhip_vqgan_encode")
        return z, loss.reshape(()), idx

    def encoder_forward(self, img):
        img = self._img(img)
        B = img.shape[0]
        x = torch.empty(B, self.tokens, self.cfg.enc.dim, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            check(self.lib.pmhip_vqgan_encoder_forward(self.handle, _p(img), B, _p(x), stream_ptr(self.device)),
                  "pmhip_vqgan_encoder_forward")
        return x

    def _new_img(self, B):
        return torch.empty(B, self.channels, self.image_size, self.image_size, device=self.device, dtype=torch.float32)

    def decode(self, z):
        z = z.to(self.device, torch.float32).contiguous()
        B = z.shape[0]
        img = self._new_img(B)
        with torch.cuda.device(self.device):
            check(self.lib.pmhip_vqgan_decode(self.handle, _p(z), B, _p(img), stream_ptr(self.device)), "pmhip_vqgan_decode")
        return img

    def decode_indices(self, idx):
        idx = idx.to(self.device, torch.int64).contiguous()
        B = idx.shape[0]
        img = self._new_img(B)
        with torch.cuda.device(self.device):
            check(self.lib.pmhip_vqgan_decode_indices(self.handle, _p(idx), B, _p(img), stream_ptr(self.device)),
                  "pmhip_vqgan_decode_indices")
        return img

    def decoder_forward(self, x):
        x = x.to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        img = self._new_img(B)
        with torch.cuda.device(self.device):
            check(self.lib.pmhip_vqgan_decoder_forward(self.handle, _p(x), B, _p(img), stream_ptr(self.device)),
                  "pmhip_vqgan_decoder_forward")
        return img


class S2Engine:
    """native CondTransformer handle (pmhip_s2) + the MaskGIT loop entry points."""

    @property
    def switches(self):
        """the PMHIP_* switches this handle latched when it was created (they are not re-read: include/pmhip.h)"""
        return _switch_dict(self.lib.pmhip_s2_switches(self.handle))

    def __init__(self, transformer, codebook, mask_token, dtype):
        self.lib = _lib.load()
        dev = next(transformer.parameters()).device
        if dev.type != "cuda":
            raise _lib.PmhipError("paintmind_amd needs the model on a ROCm device (model.to('cuda')); no CPU fallback")
        self.device, self.dtype = dev, dtype
        keep = self.keep = _Keep()
        tr = transformer
        layers = list(tr.layers)
        with torch.cuda.device(dev):
            arr, hp = _pack_tower(layers, dtype, keep, True)
            self._layer_array = arr
            w = S2Weights()
            # RAW codebook rows then the mask token (reference generate.py:148-157)
            w.tok_table = keep(torch.cat([_f32(codebook), _f32(mask_token)], dim=0).contiguous())
            w.tokproj_w, w.tokproj_b = keep(packing.pad_cols(tr.token_proj.weight, 64, dtype)), keep(_f32(tr.token_proj.bias))
            w.pos = keep(_f32(tr.position_embedding[0]))
            dim = tr.token_proj.out_features
            if isinstance(tr.context_proj, torch.nn.Linear):
                ctx_dim = tr.context_proj.in_features
                ctx_pad = round_up(ctx_dim, 64)
                w.ctxproj_w = keep(packing.pad_cols(tr.context_proj.weight, ctx_pad, dtype))
            else:
                ctx_dim = ctx_pad = dim
                w.ctxproj_w = C.c_void_p(0)
            w.layers = arr
            w.norm_g, w.norm_b = keep(_f32(tr.norm.weight)), keep(_f32(tr.norm.bias))
            w.logits_w, w.logits_b = keep(packing.cast(tr.to_logits.weight, dtype)), keep(_f32(tr.to_logits.bias))
            if dtype == torch.bfloat16:
                wg, c, d = packing.ln_fold(tr.to_logits.weight, tr.norm.weight, tr.norm.bias, dtype)
                w.logits_wf, w.logits_c, w.logits_d = keep(wg), keep(c), keep(d)
            torch.cuda.synchronize(dev)
        cfg = S2Cfg()
        cfg.tokens = tr.position_embedding.shape[1]
        cfg.embed_dim = tr.token_proj.in_features
        cfg.n_embed = tr.to_logits.out_features
        cfg.context_dim, cfg.context_dim_pad = ctx_dim, ctx_pad
        cfg.tower = TowerCfg(dim, len(layers), layers[0].attn1.heads, hp, layers[0].attn1.dim_head)
        self.cfg, self.weights = cfg, w
        self.tokens, self.n_embed, self.embed_dim, self.context_dim = cfg.tokens, cfg.n_embed, cfg.embed_dim, ctx_dim
        self.handle = C.c_void_p()
        check(self.lib.pmhip_s2_create(C.byref(self.handle), dev.index or 0, pm_dtype(dtype), C.byref(cfg), C.byref(w)),
              "pmhip_s2_create")

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.pmhip_s2_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def clone(self):
        """a second native handle (own workspace, own graphs) over the SAME packed weights: one per concurrent stream"""
        other = object.__new__(S2Engine)
        other.__dict__.update({k: v for k, v in self.__dict__.items() if k != "handle"})
        other.handle = C.c_void_p()
        check(self.lib.pmhip_s2_create(C.byref(other.handle), self.device.index or 0, pm_dtype(self.dtype), C.byref(self.cfg),
                                       C.byref(self.weights)), "pmhip_s2_create")
        return other

    def _ctx(self, context):
        if context is None:
            return None, 0
        context = context.to(self.device, torch.float32).contiguous()
        if context.shape[-1] != self.context_dim:
            raise ValueError(f"context width {context.shape[-1]} != context_dim {self.context_dim}")
        return context, context.shape[1]

    def forward(self, tokens, context=None):
        tokens = tokens.to(self.device, torch.float32).contiguous()
        B = tokens.shape[0]
        context, L = self._ctx(context)
        logits = torch.empty(B, self.tokens, self.n_embed, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            check(self.lib.pmhip_s2_forward(self.handle, _p(tokens), _p(context), L, B, _p(logits), stream_ptr(self.device)),
                  "pmhip_s2_forward")
        return logits

    def sample(self, vq_engine, ids, context, topk, temperature, num_mask, noise=None, seed=0, step=0, image_base=0,
               want_img=True, want_aux=False, guidance_scale=None):
        """one MaskGIT step; ids int64 [B,N] is updated IN PLACE (pass a clone to keep the input).
        guidance_scale (None = the reference's step): sample from uncond + scale * (cond - uncond), two tower passes."""
        B = ids.shape[0]
        context, L = self._ctx(context)
        img = vq_engine._new_img(B) if want_img else None
        pred = torch.empty(B, self.tokens, device=self.device, dtype=torch.int64) if want_aux else None
        score = torch.empty(B, self.tokens, device=self.device, dtype=torch.float32) if want_aux else None
        if noise is not None:
            noise = noise.to(self.device, torch.float32).contiguous()
        args = (self.handle, vq_engine.handle if vq_engine is not None else C.c_void_p(0), _p(ids), _p(context), L, B,
                int(topk), float(temperature), int(num_mask), _p(noise), int(seed), int(step), int(image_base), _p(img),
                _p(pred), _p(score))
        with torch.cuda.device(self.device):
            if guidance_scale is None:
                check(self.lib.pmhip_pipeline_sample(*args, stream_ptr(self.device)), "pmhip_pipeline_sample")
            else:
                check(self.lib.pmhip_pipeline_sample_guided(*args, float(guidance_scale), stream_ptr(self.device)),
                      "pmhip_pipeline_sample_guided")
        return ids, img, pred, score

    def generate(self, vq_engine, ids, context, temps, nmask, decode_flags, topk, seed=0, image_base=0, use_graph=False,
                 host=None, want_device_imgs=True, guidance_scale=None, concurrent_lanes=False):
        """T MaskGIT steps in one native call; returns imgs [n_decoded, B, C, H, W] (device) or None.

        host = (pinned float32 tensor [n_decoded, B_total, C, H, W], first row of this batch, copy stream): every decoded
        image is copied into its rows on the copy stream as soon as it is complete (the reference's `img.cpu()`,
        generate.py:195-196); the caller synchronises that stream.
        concurrent_lanes: other micro-batches run beside this call on other streams (PMHIP_GENERATE_CONCURRENT_LANES: the loop
        then does not put a small batch's decode on a side stream of its own)."""
        B = ids.shape[0]
        T = len(temps)
        context, L = self._ctx(context)
        n_dec = int(sum(1 for f in decode_flags if f))
        imgs = None
        if n_dec and (want_device_imgs or host is None):
            imgs = torch.empty(n_dec, B, vq_engine.channels, vq_engine.image_size, vq_engine.image_size, device=self.device,
                               dtype=torch.float32)
        host_ptr, host_stride, copy_stream = C.c_void_p(0), 0, C.c_void_p(0)
        if host is not None and n_dec:
            buf, row0, cstream = host
            if not (buf.is_pinned() and buf.dtype == torch.float32 and buf.is_contiguous() and buf.shape[0] == n_dec):
                raise ValueError("host image buffer must be a pinned contiguous float32 tensor [n_decoded, B_total, C, H, W]")
            per_img = buf[0, 0].numel()
            host_ptr = C.c_void_p(buf.data_ptr() + row0 * per_img * 4)
            host_stride = buf.shape[1] * per_img
            copy_stream = C.c_void_p(cstream.cuda_stream)
        temps_c = (C.c_float * T)(*[float(t) for t in temps])
        nmask_c = (C.c_int * T)(*[int(n) for n in nmask])
        dec_c = (C.c_ubyte * T)(*[1 if f else 0 for f in decode_flags])
        args = (self.handle, vq_engine.handle if vq_engine is not None else C.c_void_p(0), _p(ids), _p(context), L, B, T,
                temps_c, nmask_c, dec_c, int(topk), int(seed), int(image_base), _p(imgs), (1 if use_graph else 0) | (2 if concurrent_lanes else 0),
                stream_ptr(self.device), host_ptr, host_stride, copy_stream)
        with torch.cuda.device(self.device):
            if guidance_scale is None:
                check(self.lib.pmhip_pipeline_generate(*args), "pmhip_pipeline_generate")
            else:
                check(self.lib.pmhip_pipeline_generate_guided(*args, float(guidance_scale)), "pmhip_pipeline_generate_guided")
        return ids, imgs
