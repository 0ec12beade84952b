"""Text-conditioning producers (reference modules/encoder.py:18-104) with randomly initialised towers: output contract
(B, 77, width), device following, freeze, the layer="last"|"penultimate" rule and the checkpoint-key contract
`text_model.transformer.*` / `text_model.model.*` of a Pipeline state_dict (checkpoint row)."""
import pytest
import torch

import paintmind_amd as pm
from paintmind_amd.generate import Pipeline
from paintmind_amd.modules.encoder import CLIPTextEmbedder, T5TextEmbedder
from text_stubs import StubClipTextTower, StubTokenizer, stub_clip_tokenize, tiny_t5


def test_t5_embedder_contract():
    emb = T5TextEmbedder(tokenizer=StubTokenizer(), transformer=tiny_t5(96))
    out = emb(["a photo of a cat", "b"])
    assert out.shape == (2, 77, 96) and out.dtype == torch.float32 and not out.requires_grad
    assert all(not p.requires_grad for p in emb.parameters()) and not emb.transformer.training
    assert torch.equal(emb.encode(["b"]), out[1:2])
    assert all(k.startswith("transformer.") for k in emb.state_dict())
    # different prompts give different features; padding positions are part of the output like in the reference
    assert not torch.equal(out[0], out[1])


def test_pipeline_state_dict_carries_the_text_tower():
    """the reference's Pipeline checkpoint includes the frozen text tower under text_model.transformer.*"""
    emb = T5TextEmbedder(tokenizer=StubTokenizer(), transformer=tiny_t5(96))
    pipe = Pipeline(pm.Config(pm.ver2cfg["tiny-pipeline"]), stage1_pretrained=False, text_model=emb)
    keys = list(pipe.state_dict())
    tkeys = [k for k in keys if k.startswith("text_model.")]
    assert tkeys and all(k.startswith("text_model.transformer.") for k in tkeys)
    assert "text_model.transformer.shared.weight" in keys and "text_model.transformer.encoder.block.0.layer.0.SelfAttention.q.weight" in keys
    # round trip through a second pipeline
    emb2 = T5TextEmbedder(tokenizer=StubTokenizer(), transformer=tiny_t5(96, seed=5))
    pipe2 = Pipeline(pm.Config(pm.ver2cfg["tiny-pipeline"]), stage1_pretrained=False, text_model=emb2)
    assert not torch.equal(emb2(["x"]), emb(["x"]))
    pipe2.load_state_dict(pipe.state_dict())
    assert torch.equal(pipe2.text_model(["x"]), emb(["x"]))


def test_clip_embedder_layer_rule_on_a_stub_tower():
    tower = StubClipTextTower(width=96, layers=3)
    last = CLIPTextEmbedder(model=tower, tokenizer=stub_clip_tokenize, layer="last")
    pen = CLIPTextEmbedder(model=tower, tokenizer=stub_clip_tokenize, layer="penultimate")
    text = ["a red cube", "two"]
    a, b = last(text), pen(text)
    assert a.shape == b.shape == (2, 77, 96) and not torch.equal(a, b)
    assert all(k.startswith("model.") for k in last.state_dict())
    # manual restatement: embeddings + positions -> blocks (sequence-first, causal mask) -> ln_final
    with torch.no_grad():
        tok = stub_clip_tokenize(text)
        x = (tower.token_embedding(tok) + tower.positional_embedding).permute(1, 0, 2)
        outs = []
        for blk in tower.transformer.resblocks:
            outs.append(x)
            x = blk(x, attn_mask=tower.attn_mask)
        assert torch.allclose(a, tower.ln_final(x.permute(1, 0, 2)), atol=1e-6)
        x2 = outs[-1]                                          # input of the last block = output of the one before
        assert torch.allclose(b, tower.ln_final(x2.permute(1, 0, 2)), atol=1e-6)
    with pytest.raises(ValueError):
        CLIPTextEmbedder(model=tower, tokenizer=stub_clip_tokenize, layer="first")


def test_clip_embedder_with_open_clip_when_installed():
    open_clip = pytest.importorskip("open_clip")
    model = open_clip.create_model("ViT-B-32", pretrained=None)
    emb = CLIPTextEmbedder(model=model, tokenizer=open_clip.tokenize, layer="penultimate")
    assert emb(["a", "b"]).shape[:2] == (2, 77)


def test_clip_embedder_without_open_clip_raises_a_clear_error():
    try:
        import open_clip  # noqa: F401
        pytest.skip("open_clip is installed")
    except ImportError:
        pass
    with pytest.raises(ImportError, match="open_clip"):
        CLIPTextEmbedder()
