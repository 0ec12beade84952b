"""Pins the CPU oracle (oracle/) against the golden vectors captured from the reference itself
(tests/golden/make_goldens.py).  CPU only."""
import numpy as np
import pytest

from oracle import paintmind_oracle as O
from oracle import vq_ref
from util import load_golden, maxabs, vq_cfg, s2_cfg


@pytest.fixture(scope="module")
def tv():
    return load_golden("tiny_vqgan.npz")


@pytest.fixture(scope="module")
def tp():
    return load_golden("tiny_pipeline.npz")


def test_encoder_intermediates(tv):
    p, d = tv
    cfg = vq_cfg("tiny-vqgan")
    e = cfg["enc"]
    pe = O.patchify(d["x"], e["patch_size"]) @ p["encoder.to_patch_embedding.0.weight"].reshape(e["dim"], -1).T
    assert maxabs(pe, d["patch_embed"]) < 2e-6
    h = O.layernorm(pe + p["encoder.position_embedding"], p["encoder.norm_pre.weight"], p["encoder.norm_pre.bias"])
    assert maxabs(h, d["norm_pre"]) < 5e-6
    pre = "encoder.transformer.layers.0."
    ln = O.layernorm(h, p[pre + "norm1.weight"], p[pre + "norm1.bias"])
    assert maxabs(ln, d["l0_norm1"]) < 5e-6
    assert maxabs(O.attention(ln, None, p, pre + "attn1.", e["num_head"]), d["l0_attn1"]) < 5e-6
    collect = []
    O.encoder(d["x"], p, cfg, collect=collect)
    for i, c in enumerate(collect):
        assert maxabs(c, d[f"enc_layer{i}"]) < 2e-5


def test_vq_and_decode(tv):
    p, d = tv
    cfg = vq_cfg("tiny-vqgan")
    z, loss, idx, ze = O.vqgan_encode(d["x"], p, cfg, return_pre=True)
    assert maxabs(ze, d["prev_quant"]) < 2e-5
    assert np.array_equal(idx, d["idx"])
    assert maxabs(z, d["z"]) < 1e-6
    assert abs(float(loss) - float(d["loss"])) < 1e-6
    # distances + argmin on the golden prev_quant output (identical inputs)
    zn = O.l2norm(d["prev_quant"]).reshape(-1, 32)
    dist = O.vq_distances(zn, O.l2norm(p["quantize.embedding.weight"]))
    assert maxabs(dist, d["distances"]) < 1e-6
    assert np.array_equal(dist.argmin(1), d["idx"].reshape(-1))
    assert maxabs(O.vqgan_decode(d["z"], p, cfg), d["rec"]) < 5e-5
    x = O.linear(d["z"], p["post_quant.weight"], p["post_quant.bias"])
    assert maxabs(O.decoder(x, p, cfg), d["dec_unclamped"]) < 5e-5
    assert maxabs(O.vqgan_decode_indices(d["idx"], p, cfg), d["rec_from_idx"]) < 5e-5


def test_c_vq_oracle_matches_reference_indices(tv):
    """oracle/vq_ref.c (the arithmetic order the HIP kernel follows) vs the reference's argmin."""
    p, d = tv
    en, sq = vq_ref.prepare(p["quantize.embedding.weight"])
    idx, zn, dmin, gap = vq_ref.quantize(d["prev_quant"].reshape(-1, 32), en, sq)
    assert np.array_equal(idx, d["idx"].reshape(-1))
    assert maxabs(en, O.l2norm(p["quantize.embedding.weight"])) < 2e-7
    assert maxabs(dmin, d["distances"].min(1)) < 1e-6
    full = load_golden("full_vqgan.npz")[1]
    import torch
    torch.manual_seed(0)
    import paintmind_amd as pm
    cb = pm.create_model(arch="vqgan", version="vit-s-vqgan", pretrained=False).quantize.embedding.weight.detach().numpy()
    en, sq = vq_ref.prepare(cb)
    idx, _, _, gap = vq_ref.quantize(full["prev_quant"].reshape(-1, 32), en, sq)
    assert np.array_equal(idx, full["idx"].reshape(-1).astype(np.int64))       # 2048 tokens, 8192 codes: bit-exact
    assert maxabs(gap, full["gap"]) < 2e-6


def test_stage2_logits(tp):
    p, d = tp
    cfg = s2_cfg("tiny-pipeline")
    tok = O.ids2tokens(d["ids0"], p)
    assert np.array_equal(tok, d["tokens"])
    assert maxabs(O.cond_transformer(tok, d["context"], p, cfg), d["logits_ctx"]) < 5e-5
    assert maxabs(O.cond_transformer(tok, None, p, cfg), d["logits_noctx"]) < 5e-5


@pytest.mark.parametrize("tag", ["ctx", "noctx"])
def test_sample_step(tp, tag):
    p, d = tp
    cfg, vcfg = s2_cfg("tiny-pipeline"), vq_cfg("tiny-vqgan")
    ctx = d["context"] if tag == "ctx" else None
    B, N = d["ids0"].shape
    ids1, img1, _ = O.sample_step(d["ids0"], np.float64(0.5), ctx, 1, 1.0, np.full((B, N, 64), 0.5, np.float32), p, vcfg, cfg)
    assert np.array_equal(ids1, d[f"s1_{tag}_ids"])
    assert maxabs(img1, d[f"s1_{tag}_img"]) < 5e-5
    ids5, img5, _ = O.sample_step(d["ids0"], np.float64(0.5), ctx, 5, 0.7, d[f"s5_{tag}_noise"], p, vcfg, cfg)
    assert np.array_equal(ids5, d[f"s5_{tag}_ids"])
    assert maxabs(img5, d[f"s5_{tag}_img"]) < 5e-5


def test_decode_loop_chain(tp):
    """4 chained steps driven like Pipeline.generate (reference generate.py:189-196)."""
    p, d = tp
    cfg, vcfg = s2_cfg("tiny-pipeline"), vq_cfg("tiny-vqgan")
    B, N = d["ids0"].shape
    ids = np.full((B, N), 64, dtype=np.int64)
    T = 4
    for step in range(T):
        r = O.mask_schedule((step + 1) / T)
        ids, img, _ = O.sample_step(ids, r, d["context"], 3, 1.0 * (1 - step / T), d[f"loop_noise{step}"], p, vcfg, cfg,
                                    decode=(step == T - 1))
        assert np.array_equal(ids, d[f"loop_ids{step}"]), f"step {step}"
    assert maxabs(img, d["loop_img_last"]) < 5e-5


def test_full_size_vqgan_against_reference():
    """BASELINE config 1: vit-s-vqgan encode -> decode of 256x256 images on CPU (seed-0 weights)."""
    import torch
    import paintmind_amd as pm
    _, d = load_golden("full_vqgan.npz")
    torch.manual_seed(0)
    m = pm.create_model(arch="vqgan", version="vit-s-vqgan", pretrained=False)
    p = {k: v.numpy() for k, v in m.state_dict().items()}
    x = (torch.rand(2, 3, 256, 256, generator=torch.Generator().manual_seed(100)) * 2 - 1).numpy()
    cfg = vq_cfg("vit-s-vqgan")
    z, loss, idx, ze = O.vqgan_encode(x[:1], p, cfg, return_pre=True)
    assert maxabs(ze, d["prev_quant"][:1]) < 1e-4
    mism = idx[0] != d["idx"][0]
    # any disagreement must be a near-tie of the reference's own distances
    assert mism.sum() <= 2 and np.all(d["gap"][:1024][mism] < 1e-5), (int(mism.sum()), d["gap"][:1024][mism])
    rec = O.vqgan_decode(d["z"][:1], p, cfg)
    assert maxabs(rec[:, :, ::4, ::4], d["rec_sub"][:1]) < 1e-3
    assert abs(float(loss) - float(d["loss"])) < 1e-3


def test_philox_reference_vector():
    """Philox4x32-10 known-answer tests (Random123 kat_vectors: zero and all-ones counter/key)."""
    out = O.philox4x32_10(0, 0, 0, 0, 0, 0)
    assert [int(v) for v in out] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    out = O.philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)
    assert [int(v) for v in out] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    u = O.philox_uniform(5, 3, np.arange(1000), np.arange(1000) % 64)
    assert u.min() >= 0 and u.max() < 1 and abs(float(u.mean()) - 0.5) < 0.05


def test_torch_port_matches_goldens(tv, tp):
    """oracle/torch_port.py (what bench.py times as cpu_baseline) against the same reference vectors."""
    import torch
    from oracle import torch_port as TP
    p, d = tv
    pt = {k: torch.from_numpy(v) for k, v in p.items()}
    cfg = vq_cfg("tiny-vqgan")
    with torch.no_grad():
        z, loss, idx = TP.vqgan_encode(torch.from_numpy(d["x"]), pt, cfg)
        assert np.array_equal(idx.numpy(), d["idx"]) and maxabs(z.numpy(), d["z"]) < 1e-6
        assert maxabs(TP.vqgan_decode(torch.from_numpy(d["z"]), pt, cfg).numpy(), d["rec"]) < 5e-5
        p2, d2 = tp
        pt2 = {k: torch.from_numpy(v) for k, v in p2.items()}
        ids, img, logits = TP.sample_step(torch.from_numpy(d2["ids0"]), 0.5, torch.from_numpy(d2["context"]), 5, 0.7,
                                          torch.from_numpy(d2["s5_ctx_noise"]), pt2, cfg, s2_cfg("tiny-pipeline"))
        assert maxabs(logits.numpy(), d2["logits_ctx"]) < 5e-5
        assert np.array_equal(ids.numpy(), d2["s5_ctx_ids"]) and maxabs(img.numpy(), d2["s5_ctx_img"]) < 5e-5


@pytest.mark.parametrize("key,name", [("bench-text-24L-d768", "full_stage2_d768"), ("bench-text-24L-d1024-512px", "full_stage2_d1024")])
def test_north_star_size_models_weights_and_oracle_against_the_reference(key, name):
    """The model north_star's target is quoted on (24L/d768, 77 x 768 context, context_proj = Identity) and cfg 5's stage 2
    (24L/d1024, context_proj 768 -> 1024), against the REFERENCE's own output at that size (make_goldens.py full_stage2_text):
    the seeded init reproduces the reference's weights bit for bit (sha256), and the torch port of the oracle -- what the GPU
    tests of these configs compare with at larger batches -- reproduces its logits."""
    import hashlib
    import torch
    import paintmind_amd as pm
    from paintmind_amd.generate import Pipeline
    from oracle import torch_port as TP
    from util import api_facts
    _, d = load_golden(name + ".npz")
    torch.manual_seed(0)
    pipe = Pipeline(pm.Config(pm.ver2cfg[key]), stage1_pretrained=False).eval()
    h = hashlib.sha256()
    sd = {k: v for k, v in pipe.state_dict().items() if not k.startswith("text_model")}
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.numpy().tobytes())
    assert h.hexdigest() == api_facts()[name + "_weights_sha256"] == bytes(d["weights_sha"]).hex()
    ids0 = torch.from_numpy(d["ids0"].astype(np.int64))
    table = torch.cat([sd["vqgan.quantize.embedding.weight"], sd["mask_token"]])
    with torch.no_grad():
        logits = TP.cond_transformer(table[ids0], torch.from_numpy(d["context"]), sd, s2_cfg(key))
    assert maxabs(logits[:, ::8, ::64].numpy(), d["logits_sub"]) < 1e-4
    assert maxabs(torch.logsumexp(logits, -1).numpy(), d["logits_lse"]) < 1e-4
    bad = logits.argmax(-1).numpy() != d["logits_argmax"]
    assert np.all(d["logits_top2gap"][bad] < 1e-4)


def test_vit_b_512_weights_and_oracle_against_the_reference():
    """cfg 5's assumed stage 1 (vit-b-vqgan-512): seeded weights sha-identical to the reference's, and the torch port of the oracle
    reproduces the reference's tokens and reconstruction (tests/golden/full_vqgan_b512.npz)."""
    import hashlib
    import torch
    import paintmind_amd as pm
    from oracle import torch_port as TP
    from util import api_facts
    _, d = load_golden("full_vqgan_b512.npz")
    torch.manual_seed(0)
    m = pm.create_model(arch="vqgan", version="vit-b-vqgan-512", pretrained=False).eval()
    h = hashlib.sha256()
    for k, v in m.state_dict().items():
        h.update(k.encode())
        h.update(v.numpy().tobytes())
    assert h.hexdigest() == api_facts()["full_vqgan_b512_weights_sha256"] == bytes(d["weights_sha"]).hex()
    sd = dict(m.state_dict())
    x = torch.rand(1, 3, 512, 512, generator=torch.Generator().manual_seed(101)) * 2 - 1
    with torch.no_grad():
        z, loss, idx = TP.vqgan_encode(x, sd, vq_cfg("vit-b-vqgan-512"))
        mism = idx.numpy().reshape(-1) != d["idx"].reshape(-1)
        assert mism.sum() <= 2 and np.all(d["gap"][mism] < 1e-5)
        rec = TP.vqgan_decode(torch.from_numpy(d["z"]), sd, vq_cfg("vit-b-vqgan-512"))
    assert maxabs(rec.numpy()[:, :, ::8, ::8], d["rec_sub"]) < 1e-4


# ---- masked-token objective, forward only (generate.py:78-146) ---------------------------------------
@pytest.fixture(scope="module")
def tf():
    return load_golden("tiny_forward.npz")[1]


def test_random_masking_golden(tp, tf):
    p, _ = tp
    for i in range(4):
        x, mask = O.random_masking(tf["rm_x"], p["mask_token"], float(tf[f"rm{i}_ratio"]), tf[f"rm{i}_noise"])
        assert np.array_equal(mask, tf[f"rm{i}_mask"])
        assert np.array_equal(x, tf[f"rm{i}_x"])
    assert tf["rm2_mask"].sum(1).tolist() == [1, 1, 1]            # ratio 0.01 -> max(int(L*r),1) = 1 masked
    assert tf["rm3_mask"].min() == 1                              # ratio 1.0 -> everything masked


def test_masked_cross_entropy_golden(tf):
    loss, rows = O.masked_ce(tf["ce_logit"], tf["ce_label"], tf["ce_mask"], 0.1)
    assert abs(float(loss) - float(tf["ce_loss"])) < 1e-5
    assert maxabs(rows, tf["ce_rows"] * tf["ce_mask"].reshape(-1)) < 1e-5


def test_pipeline_forward_loss_golden(tp, tf):
    p, _ = tp
    cfg, s2 = vq_cfg("tiny-vqgan"), s2_cfg("tiny-pipeline")
    for tag, ctx in (("ctx", tf["context"]), ("noctx", None)):
        for j in range(2):
            loss, _ = O.pipeline_forward(tf["img"], ctx, float(tf[f"fw_{tag}{j}_ratio"]), tf[f"fw_{tag}{j}_noise"], p, cfg, s2)
            assert abs(float(loss) - float(tf[f"fw_{tag}{j}_loss"])) < 2e-5
