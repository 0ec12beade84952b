"""Text-conditioning producers (reference modules/encoder.py:18-42): the frozen towers are third-party
pretrained models that stay on stock PyTorch-ROCm; the hot path only consumes their (B,77,ctx) output.
"""
import torch
import torch.nn as nn


class T5TextEmbedder(nn.Module):
    """Flan-T5 encoder wrapper; unlike the reference the device follows the module (no hard-coded "cuda")."""

    def __init__(self, version="google/flan-t5-xl", device=None, max_length=77, freeze=True):
        super().__init__()
        from transformers import T5EncoderModel, T5Tokenizer   # needs local weights: there is no network here
        self.tokenizer = T5Tokenizer.from_pretrained(version)
        self.transformer = T5EncoderModel.from_pretrained(version)
        self.max_length = max_length
        if freeze:
            self.freeze()

    def freeze(self):
        self.transformer = self.transformer.eval()
        for param in self.parameters():
            param.requires_grad = False

    def forward(self, text):
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        tokens = enc["input_ids"].to(next(self.transformer.parameters()).device)
        return self.transformer(input_ids=tokens).last_hidden_state

    def encode(self, text):
        return self(text)


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
