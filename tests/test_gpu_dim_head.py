"""dim_head other than 64 (reference modules/attention.py:27-33 accepts any; every shipped config uses 64).

The tuned MFMA kernels serve 64; 16, 32, ..., 128 run the plain path behind pmhip_gemm_heads_dh / pmhip_attention_dh.
Checked at the operator level against a torch fp32 restatement, and at the model level (operator composition AND the
fused C++ engine) against the numpy oracle on the same seeded weights.
"""
import numpy as np
import pytest
import torch

import paintmind_amd as pm
from gpu_common import dev, n, t
from oracle import paintmind_oracle as O
from paintmind_amd import ops
from paintmind_amd._lib import PmhipError
from paintmind_amd.modules.attention import CrossAttention
from paintmind_amd.modules.encoder import SyntheticTextEmbedder
from paintmind_amd.config import _vqgan, ver2cfg
from paintmind_amd.generate import Pipeline
from util import maxabs

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _ref_attention(q, k, v, scale):
    """q [B,H,Nq,dh], k/v [B,H,Nkv,dh], fp32: softmax(q k^T * scale) v -> [B*Nq, H*dh]  (attention.py:51-58)"""
    sim = torch.einsum("bhid,bhjd->bhij", q.double() * scale, k.double())
    o = torch.einsum("bhij,bhjd->bhid", sim.softmax(-1), v.double())
    B, H, Nq, dh = q.shape
    return o.permute(0, 2, 1, 3).reshape(B * Nq, H * dh).float()


@pytest.mark.parametrize("dh,heads", [(16, 4), (32, 2), (48, 4), (96, 2), (128, 1), (128, 3)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_projection_and_attention_any_dim_head(dh, heads, dtype):
    if (heads * dh) % 64:
        pytest.skip("heads*dim_head must be a multiple of 64")
    g = torch.Generator().manual_seed(dh * 7 + heads)
    B, Nq, Nkv, D = 3, 50, 77, 128                         # ragged: neither length is a multiple of 64
    inner = heads * dh
    x = torch.randn(B * Nq, D, generator=g)
    c = torch.randn(B * Nkv, D, generator=g)
    wq = torch.randn(inner, D, generator=g) * D ** -0.5
    wkv = torch.randn(2 * inner, D, generator=g) * D ** -0.5
    scale = dh ** -0.5
    fast = dtype == torch.bfloat16
    rd = (lambda a: a.to(dtype).float()) if fast else (lambda a: a)
    (q,) = ops.gemm_heads(x.to(dev(), dtype), wq.to(dev(), dtype), heads, Nq, [ops.PART_Q], scale * (ops.LOG2E if fast else 1.0), dh)
    k, vt = ops.gemm_heads(c.to(dev(), dtype), wkv.to(dev(), dtype), heads, Nkv, [ops.PART_K, ops.PART_V], 1.0, dh)
    assert q.shape == (B, heads, Nq, dh) and k.shape == (B, heads, 128, dh) and vt.shape == (B, heads, dh, 128)
    # the projections themselves
    q_ref = (rd(x) @ rd(wq).T).reshape(B, Nq, heads, dh).permute(0, 2, 1, 3)
    kv_ref = (rd(c) @ rd(wkv).T).reshape(B, Nkv, 2, heads, dh)
    k_ref, v_ref = kv_ref[:, :, 0].permute(0, 2, 1, 3), kv_ref[:, :, 1].permute(0, 2, 1, 3)
    ptol = 2e-2 if fast else 1e-4
    assert maxabs(n(q) / (ops.LOG2E if fast else 1.0), n(q_ref * scale)) < ptol
    assert maxabs(n(k[:, :, :Nkv]), n(k_ref)) < ptol and maxabs(n(vt[:, :, :, :Nkv].transpose(2, 3)), n(v_ref)) < ptol
    assert float(k[:, :, Nkv:].abs().max()) == 0 and float(vt[:, :, :, Nkv:].abs().max()) == 0     # padding untouched
    # attention on exactly the operands the kernel saw
    o = ops.attention(q, k, vt, Nkv, use_exp2=fast)
    assert o.shape == (B * Nq, inner)
    qf = q.float().cpu() / (ops.LOG2E if fast else 1.0)
    ref = _ref_attention(qf, k[:, :, :Nkv].float().cpu(), vt[:, :, :, :Nkv].transpose(2, 3).float().cpu(), 1.0)
    assert maxabs(n(o), n(ref)) < (2e-2 if fast else 1e-5)


def test_unsupported_dim_head_is_refused():
    with pytest.raises(ValueError):
        CrossAttention(64, heads=2, dim_head=24)
    with pytest.raises(ValueError):
        CrossAttention(64, heads=1, dim_head=32)        # inner 32 is not a multiple of 64
    q = torch.zeros(1, 1, 8, 24, device=dev())
    with pytest.raises(PmhipError):
        ops.attention(q, torch.zeros(1, 1, 64, 24, device=dev()), torch.zeros(1, 1, 24, 64, device=dev()), 8)


@pytest.fixture(scope="module")
def dh_models():
    """a ViT-VQGAN with dim_head 32 and a text-conditional stage 2 with dim_head 128, seeded; numpy copies of the weights"""
    vq = _vqgan(32, 8, 64, 2, 4, 128, n_embed=64, embed_dim=32)
    for tower in ("enc", "dec"):
        vq[tower]["dim_head"] = 32
    ver2cfg["dh32-vqgan"] = vq
    ver2cfg["dh128-pipeline"] = {"stage1": "dh32-vqgan", "t5": "t5-l", "text_model": "none", "context_dim": 96,
                                 "dim": 128, "dim_head": 128, "mlp_dim": 256, "num_head": 1, "depth": 2, "dropout": 0.1}
    torch.manual_seed(11)
    pipe = Pipeline(pm.Config(ver2cfg["dh128-pipeline"]), stage1_pretrained=False).eval()
    p = {k: v.detach().numpy().copy() for k, v in pipe.state_dict().items() if not k.startswith("text_model")}
    yield pipe.to(dev()), p, ver2cfg["dh128-pipeline"], vq
    del ver2cfg["dh32-vqgan"], ver2cfg["dh128-pipeline"]


def test_models_with_other_dim_head_match_oracle(dh_models):
    pipe, p, scfg, vcfg = dh_models
    assert pipe.vqgan.encoder.transformer.layers[0].attn1.dim_head == 32
    assert pipe.transformer.layers[0].attn1.dim_head == 128 and pipe.transformer.layers[0].attn1.to_q.out_features == 128
    vq_p = {k[len("vqgan."):]: v for k, v in p.items() if k.startswith("vqgan.")}
    g = torch.Generator().manual_seed(2)
    x = torch.rand(3, 3, 32, 32, generator=g) * 2 - 1
    z_ref, loss_ref, idx_ref = O.vqgan_encode(x.numpy(), vq_p, vcfg)
    z, loss, idx = pipe.vqgan.encode(x.to(dev()))
    assert np.array_equal(n(idx), idx_ref) and maxabs(n(z), z_ref) < 1e-5
    assert maxabs(n(pipe.vqgan.decode_from_indice(idx)), O.vqgan_decode_indices(idx_ref, vq_p, vcfg)) < TOL
    # operator composition == fused engine
    assert torch.equal(pipe.vqgan.encoder(x.to(dev())), pipe.vqgan.engine().encoder_forward(x.to(dev())))
    V, N = 64, pipe.num_tokens
    ids0 = torch.randint(0, V, (3, N), generator=g)
    ids0[torch.rand(3, N, generator=g) < 0.6] = V
    ctx = torch.randn(3, 77, 96, generator=g)
    tok = pipe.ids2tokens(ids0.to(dev()))
    for c_gpu, c_np in ((ctx.to(dev()), ctx.numpy()), (None, None)):
        ref = O.cond_transformer(O.ids2tokens(ids0.numpy(), p), c_np, p, scfg)
        assert maxabs(n(pipe.tokens2logits(tok, c_gpu)), ref) < TOL            # engine
        assert maxabs(n(pipe.transformer(tok, c_gpu)), ref) < TOL              # operators


def test_decode_loop_with_other_dim_head(dh_models):
    """whole loop (graph replay + lanes) runs, is deterministic, and its first step equals the oracle's step"""
    pipe, p, scfg, vcfg = dh_models
    prompts = [f"p{i}" for i in range(5)]
    assert isinstance(pipe.text_model, SyntheticTextEmbedder)
    a = pipe.generate(prompts, timesteps=4, topk=3, temperature=1.0, seed=9)[-1]
    b = pipe.generate(prompts, timesteps=4, topk=3, temperature=1.0, seed=9)[-1]
    assert a.shape == (5, 3, 32, 32) and torch.equal(a, b) and torch.isfinite(a).all()
    ctx = pipe.text_model(prompts)
    ids0 = torch.full((5, pipe.num_tokens), 64, dtype=torch.long, device=dev())
    noise = torch.rand(5, pipe.num_tokens, 64, generator=torch.Generator().manual_seed(1)).to(dev())
    ids1, _ = pipe.sample(ids0, np.float64(0.5), text=ctx, topk=3, temperature=0.8, noise=noise)
    vq_p = {k[len("vqgan."):]: v for k, v in p.items() if k.startswith("vqgan.")}
    ids_ref, _, _ = O.sample_step(ids0.cpu().numpy(), 0.5, ctx.cpu().numpy(), 3, 0.8, noise.cpu().numpy(), p, vcfg, scfg, decode=False)
    assert np.array_equal(n(ids1), ids_ref)


def test_bf16_mode_with_other_dim_head(dh_models):
    pipe, p, scfg, vcfg = dh_models
    g = torch.Generator().manual_seed(4)
    ids0 = torch.randint(0, 64, (2, pipe.num_tokens), generator=g).to(dev())
    ctx = torch.randn(2, 77, 96, generator=g).to(dev())
    tok = pipe.ids2tokens(ids0)
    ref = pipe.tokens2logits(tok, ctx)
    pipe.set_compute_dtype(torch.bfloat16)
    try:
        got = pipe.tokens2logits(tok, ctx)
    finally:
        pipe.set_compute_dtype(torch.float32)
    assert got.dtype == torch.float32 and maxabs(n(got), n(ref)) < 0.05
    cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0)
    assert float(cos) > 0.9995
