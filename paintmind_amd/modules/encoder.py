"""Text-conditioning producers (reference modules/encoder.py:18-104): the frozen towers are third-party
pretrained models that stay on stock PyTorch-ROCm; the hot path only consumes their (B,77,ctx) output.

Both wrappers keep the reference's attribute names (`transformer` for the T5 encoder, `model` for the CLIP tower),
because those names are the checkpoint-key contract: a Pipeline state_dict carries `text_model.transformer.*`
(T5) or `text_model.model.*` (CLIP).  Unlike the reference, the device follows the module (`.to()`), there is no
hard-coded "cuda" (reference encoder.py:19,36,57,80), and a pre-built tower / tokenizer can be injected, which is how
the tests run them with randomly initialised weights (there is no network here).
"""
import torch
import torch.nn as nn

CLIP_ARCH = 'ViT-L-14'                  # reference encoder.py:14-15
CLIP_VERSION = 'laion2b_s32b_b82k'


def _module_device(module):
    return next(module.parameters()).device


class T5TextEmbedder(nn.Module):
    """Flan-T5 encoder (reference encoder.py:18-42): tokens padded / truncated to max_length -> last_hidden_state."""

    def __init__(self, version="google/flan-t5-xl", device=None, max_length=77, freeze=True, tokenizer=None, transformer=None):
        super().__init__()
        if tokenizer is None or transformer is None:
            from transformers import T5EncoderModel, T5Tokenizer   # needs local weights: there is no network here
            tokenizer = tokenizer or T5Tokenizer.from_pretrained(version)
            transformer = transformer or T5EncoderModel.from_pretrained(version)
        self.tokenizer = tokenizer
        self.transformer = transformer
        self.max_length = max_length
        if device is not None:
            self.to(device)
        if freeze:
            self.freeze()

    def freeze(self):
        self.transformer = self.transformer.eval()
        for param in self.parameters():
            param.requires_grad = False

    @torch.no_grad()
    def forward(self, text):
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        tokens = enc["input_ids"].to(_module_device(self.transformer))
        return self.transformer(input_ids=tokens).last_hidden_state

    def encode(self, text):
        return self(text)


class CLIPTextEmbedder(nn.Module):
    """OpenCLIP text tower (reference encoder.py:45-104).  layer="last": every residual block, then ln_final;
    layer="penultimate": the last block is skipped (ln_final still applied).  Output (B, 77, width), all positions.

    Written against the open_clip model interface: `token_embedding`, `positional_embedding`, `transformer.resblocks`
    (each block called as block(x, attn_mask=...) on sequence-first tensors), `attn_mask`, `ln_final`."""
    LAYERS = ("last", "penultimate")

    def __init__(self, arch=CLIP_ARCH, version=CLIP_VERSION, device=None, max_length=77, layer="last", precision='fp32',
                 freeze=True, model=None, tokenizer=None):
        super().__init__()
        if layer not in self.LAYERS:
            raise ValueError(f"layer must be one of {self.LAYERS}, got {layer!r}")
        if model is None or tokenizer is None:
            try:
                import open_clip
            except ImportError as e:                    # the tower is optional: only this class needs the package
                raise ImportError("CLIPTextEmbedder needs the open_clip package (or pass model= and tokenizer=)") from e
            if model is None:
                model = open_clip.create_model(arch, pretrained=version, precision=precision)
                if hasattr(model, "visual"):
                    del model.visual                    # the image tower is never used here
            tokenizer = tokenizer or open_clip.tokenize
        self.model = model
        self.tokenize = tokenizer
        self.max_length = max_length
        self.layer = layer
        self.skip_last = {"last": 0, "penultimate": 1}[layer]
        if device is not None:
            self.to(device)
        if freeze:
            self.freeze()

    def freeze(self):
        self.model = self.model.eval()
        for param in self.parameters():
            param.requires_grad = False

    @torch.no_grad()
    def forward(self, text):
        tokens = self.tokenize(text)
        return self.encode_with_transformer(tokens.to(_module_device(self.model)))

    def encode_with_transformer(self, tokens):
        m = self.model
        x = m.token_embedding(tokens) + m.positional_embedding          # (B, n_ctx, width)
        blocks = list(m.transformer.resblocks)
        blocks = blocks[:len(blocks) - self.skip_last]
        mask = getattr(m, "attn_mask", None)
        seq_first = not getattr(m.transformer, "batch_first", False)   # older open_clip blocks are sequence-first
        if seq_first:
            x = x.permute(1, 0, 2)
        for blk in blocks:
            x = blk(x, attn_mask=mask)
        if seq_first:
            x = x.permute(1, 0, 2)
        return m.ln_final(x)

    def encode(self, text):
        return self(text)


class NullTextEmbedder(nn.Module):
    """Unconditional generation through the drop-in API: the context is None, so attn2 runs as a second self-attention
    (reference modules/attention.py:47).  BASELINE.json configs[2] is this path."""

    def forward(self, text):
        return None


class SyntheticTextEmbedder(nn.Module):
    """Deterministic stand-in used by the benches and tests: N(0,1) features keyed by (seed, prompt index).

    Row i of the output depends only on (seed, base_index + i), so sharding a prompt list over ranks
    yields the same features as a single process.
    """

    def __init__(self, context_dim, max_length=77, seed=1234):
        super().__init__()
        self.context_dim, self.max_length, self.seed = context_dim, max_length, seed
        self.base_index = 0
        self.register_buffer("_anchor", torch.zeros(1), persistent=False)

    def forward(self, text):
        rows = []
        for i in range(len(text)):
            g = torch.Generator().manual_seed(self.seed * 1000003 + self.base_index + i)
            rows.append(torch.randn(self.max_length, self.context_dim, generator=g))
        return torch.stack(rows).to(self._anchor.device)
