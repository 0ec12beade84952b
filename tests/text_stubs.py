"""Randomly initialised stand-ins for the third-party text towers (no network, no pretrained weights): a tiny real
T5EncoderModel with a deterministic tokenizer, and a tower with the open_clip text-model attribute contract."""
import torch
import torch.nn as nn


class StubTokenizer:
    """same call contract as T5Tokenizer(text, truncation, max_length, padding='max_length', return_tensors='pt')"""

    def __init__(self, vocab=128):
        self.vocab = vocab

    def __call__(self, text, max_length=77, **kw):
        ids = torch.zeros(len(text), max_length, dtype=torch.long)
        for i, s in enumerate(text):
            codes = [1 + (ord(c) % (self.vocab - 2)) for c in s][:max_length - 1] + [1]
            ids[i, :len(codes)] = torch.tensor(codes)
        return {"input_ids": ids}


def tiny_t5(d_model=96, seed=0):
    from transformers import T5Config, T5EncoderModel
    torch.manual_seed(seed)
    return T5EncoderModel(T5Config(vocab_size=128, d_model=d_model, d_kv=16, d_ff=128, num_layers=2, num_heads=4))


class _Block(nn.Module):
    def __init__(self, width, heads):
        super().__init__()
        self.ln_1, self.ln_2 = nn.LayerNorm(width), nn.LayerNorm(width)
        self.attn = nn.MultiheadAttention(width, heads)
        self.mlp = nn.Sequential(nn.Linear(width, 2 * width), nn.GELU(), nn.Linear(2 * width, width))

    def forward(self, x, attn_mask=None):                      # sequence-first, like open_clip's ResidualAttentionBlock
        h = self.ln_1(x)
        x = x + self.attn(h, h, h, need_weights=False, attn_mask=attn_mask)[0]
        return x + self.mlp(self.ln_2(x))


class _Transformer(nn.Module):
    def __init__(self, width, layers, heads):
        super().__init__()
        self.resblocks = nn.ModuleList([_Block(width, heads) for _ in range(layers)])
        self.grad_checkpointing = False


class StubClipTextTower(nn.Module):
    """attribute contract of open_clip's CLIP text side: token_embedding, positional_embedding, transformer.resblocks,
    attn_mask (causal), ln_final"""

    def __init__(self, width=96, layers=3, heads=4, n_ctx=77, vocab=256, seed=0):
        super().__init__()
        torch.manual_seed(seed)
        self.token_embedding = nn.Embedding(vocab, width)
        self.positional_embedding = nn.Parameter(0.01 * torch.randn(n_ctx, width))
        self.transformer = _Transformer(width, layers, heads)
        self.ln_final = nn.LayerNorm(width)
        self.register_buffer("attn_mask", torch.full((n_ctx, n_ctx), float("-inf")).triu_(1), persistent=False)


def stub_clip_tokenize(text, n_ctx=77, vocab=256):
    ids = torch.zeros(len(text), n_ctx, dtype=torch.long)
    for i, s in enumerate(text):
        codes = [2 + (ord(c) % (vocab - 3)) for c in s][:n_ctx - 1] + [1]
        ids[i, :len(codes)] = torch.tensor(codes)
    return ids
