"""Import the upstream reference (PUBLIC, read-only at /root/reference) in THIS container only.

The reference's package __init__ pulls torchvision/lpips/kornia/open_clip, which are absent here
(SURVEY.md section 8(c)).  We pre-seed sys.modules with three stub modules so that the hot-path
files (stage1/*, stage2/*, modules/attention.py, modules/mlp.py, generate.py, config.py,
factory.py) import unchanged.  Nothing from /root/reference is copied: this file only makes the
reference importable so that make_goldens.py can record input/output vectors.

Never imported by the product or by tests that run on the GPU box.
"""
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = "/root/reference"


class StubTextEmbedder(nn.Module):
    """Stands in for the frozen T5 tower: returns a seeded (B,77,ctx_dim) tensor."""
    ctx_dim = 1024
    seed = 1234

    def __init__(self, version=None, device="cpu", max_length=77, freeze=True):
        super().__init__()
        self.max_length = max_length

    def forward(self, text):
        g = torch.Generator().manual_seed(self.seed)
        return torch.randn(len(text), self.max_length, self.ctx_dim, generator=g)


def import_reference():
    if "paintmind" in sys.modules and getattr(sys.modules["paintmind"], "__file__", "").startswith(REF_ROOT):
        return sys.modules["paintmind"]
    trainer = types.ModuleType("paintmind.utils.trainer")
    trainer.VQGANTrainer = object
    trainer.PaintMindTrainer = object
    transform = types.ModuleType("paintmind.utils.transform")
    transform.stage1_transform = lambda *a, **k: None
    transform.stage2_transform = lambda *a, **k: None
    encoder = types.ModuleType("paintmind.modules.encoder")
    encoder.T5TextEmbedder = StubTextEmbedder
    sys.modules["paintmind.utils.trainer"] = trainer
    sys.modules["paintmind.utils.transform"] = transform
    sys.modules["paintmind.modules.encoder"] = encoder
    sys.path.insert(0, REF_ROOT)
    import paintmind  # noqa
    assert paintmind.__file__.startswith(REF_ROOT)
    from paintmind.modules import attention
    assert attention.XFORMERS_IS_AVAILBLE is False
    return paintmind
