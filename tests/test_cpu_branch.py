"""The module tree on the CPU (BASELINE.json configs[0]: `pm.create_model` + `model.encode/decode` on CPU, no GPU).

The reference runs on whatever device its modules live on (stage1/vqmodel.py:21-30, generate.py:159-198).  So does this
package: with the parameters on the CPU every module takes its own plain-torch branch (paintmind_amd/modules/*.py,
stage1/*.py, stage2/transformer.py, generate.py -- product code, nothing from oracle/), fp32; on a ROCm device the same
calls go to the HIP engine.  Checked here against the vectors captured from the reference (tests/golden/make_goldens.py)."""
import numpy as np
import torch

import paintmind_amd as pm
from paintmind_amd.generate import Pipeline
from util import api_facts, load_golden, maxabs, to_torch_sd


def test_config1_vit_s_vqgan_encode_decode_on_cpu():
    _, d = load_golden("full_vqgan.npz")
    torch.manual_seed(0)
    m = pm.create_model(arch="vqgan", version="vit-s-vqgan", pretrained=False).eval()
    x = torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(100)) * 2 - 1
    z, loss, idx = m.encode(x[:1])
    assert z.shape == (1, 1024, 32) and idx.dtype == torch.int64 and loss.shape == ()
    mism = idx[0].numpy() != d["idx"][0]
    assert mism.sum() <= 2 and np.all(d["gap"][:1024][mism] < 1e-5)          # token indices: exact up to reference near-ties
    assert maxabs(z.numpy(), d["z"][:1]) < 1e-5 or mism.any()
    rec = m.decode(torch.from_numpy(d["z"][:1]))
    assert rec.shape == (1, 3, 256, 256) and float(rec.abs().max()) <= 1.0
    assert maxabs(rec.numpy()[:, :, ::4, ::4], d["rec_sub"][:1]) < 1e-3
    assert abs(float(loss) - float(d["loss"])) < 1e-3
    rec2, loss2 = m(x[:1])                                                   # VQModel.forward -> (rec, loss)
    assert rec2.shape == rec.shape and float((m.decode_from_indice(idx) - rec2).abs().max()) < 1e-4


def test_tiny_vqgan_and_pipeline_goldens_on_cpu():
    p, d = load_golden("tiny_vqgan.npz")
    m = pm.create_model(arch="vqgan", version="tiny-vqgan", pretrained=False).eval()
    m.load_state_dict(to_torch_sd(p))
    z, loss, idx = m.encode(torch.from_numpy(d["x"]))
    assert np.array_equal(idx.numpy(), d["idx"]) and maxabs(z.numpy(), d["z"]) < 1e-5
    assert maxabs(m.decode(z).numpy(), d["rec"]) < 1e-4
    assert maxabs(m.decode_from_indice(idx).numpy(), d["rec_from_idx"]) < 1e-4

    p, d = load_golden("tiny_pipeline.npz")
    pipe = Pipeline(pm.Config(pm.ver2cfg["tiny-pipeline"]), stage1_pretrained=False).eval()
    missing = pipe.load_state_dict(to_torch_sd(p), strict=False)
    assert not [k for k in missing.missing_keys if not k.startswith("text_model")]
    ids0, ctx = torch.from_numpy(d["ids0"]), torch.from_numpy(d["context"])
    tok = pipe.ids2tokens(ids0)
    assert np.array_equal(tok.numpy(), d["tokens"])
    assert maxabs(pipe.tokens2logits(tok, ctx).detach().numpy(), d["logits_ctx"]) < 1e-4      # (autograd-capable, like the reference)
    assert maxabs(pipe.tokens2logits(tok, None).detach().numpy(), d["logits_noctx"]) < 1e-4
    for tag, c in (("ctx", ctx), ("noctx", None)):                           # one MaskGIT step with the reference's captured noise
        ids5, img5 = pipe.sample(ids0, np.float64(0.5), text=c, topk=5, temperature=0.7, noise=torch.from_numpy(d[f"s5_{tag}_noise"]))
        assert np.array_equal(ids5.numpy(), d[f"s5_{tag}_ids"]) and maxabs(img5.numpy(), d[f"s5_{tag}_img"]) < 1e-4
    facts = api_facts()
    imgs = pipe.generate(["a", "b"], timesteps=8, temperature=1.0, topk=5, save_interval=2, seed=5)
    assert len(imgs) == facts["generate_T8_si2_len"] and list(imgs[0].shape) == facts["generate_img_shape"]
    imgs2, ids2 = pipe.generate(["a", "b"], timesteps=8, topk=5, save_interval=2, seed=5, return_ids=True)
    assert all(torch.equal(a, b) for a, b in zip(imgs, imgs2))
    assert int((ids2 == 64).sum(1).min()) == facts["residual_mask_tokens_after_T8"]

    # guided step (extension) on the CPU branch against the oracle's restatement
    from oracle import paintmind_oracle as O
    g_ids, g_img = pipe.sample(ids0, np.float64(0.5), text=ctx, topk=5, temperature=0.7, noise=torch.from_numpy(d["s5_ctx_noise"]),
                               guidance_scale=2.5)
    o_ids, o_img, _ = O.sample_step(d["ids0"], np.float64(0.5), d["context"], 5, 0.7, d["s5_ctx_noise"], p, pm.ver2cfg["tiny-vqgan"],
                                    pm.ver2cfg["tiny-pipeline"], guidance_scale=2.5)
    assert np.array_equal(g_ids.numpy(), o_ids) and maxabs(g_img.numpy(), o_img) < 1e-4
    g = pipe.generate(["a", "b"], timesteps=4, topk=3, save_interval=2, seed=5, guidance_scale=2.5)
    assert len(g) == 2 and list(g[0].shape) == facts["generate_img_shape"]
