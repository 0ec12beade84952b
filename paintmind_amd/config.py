"""Attribute-bag configuration + named presets.

Behavioural mirror of the reference's ``Config`` (reference paintmind/config.py:4-37) and of its two
named presets (config.py:40-82).  The preset *values* are the interchange contract for checkpoints
(state_dict shapes derive from them) so they are restated exactly; the extra ``bench-*`` presets
are the synthetic configurations BASELINE.json names expressed with the
same keys.
"""
import copy
import json


class Config:
    """Dict-backed attribute bag with JSON (de)serialisation (reference config.py:4-37)."""

    def __init__(self, config=None):
        if config is not None:
            self.from_dict(config)

    def __repr__(self):
        return str(self.to_json_string())

    def to_dict(self):
        return copy.deepcopy(self.__dict__)

    def to_json_string(self):
        return json.dumps(self.to_dict(), indent=2)

    def to_json(self, path):
        with open(path, "w") as fh:
            json.dump(self.to_dict(), fh, indent=2)

    def from_dict(self, dct):
        self.clear()
        for key, value in dct.items():
            self.__dict__[key] = value
        return self.to_dict()

    def from_json(self, json_path):
        with open(json_path, "r") as fh:
            self.from_dict(json.load(fh))
        return self.to_dict()

    def clear(self):
        # the reference does ``del self.__dict__`` (config.py:36-37); the observable effect is an
        # empty attribute bag
        self.__dict__.clear()


def _tower(image_size, patch_size, dim, depth, num_head, mlp_dim, channel_key, dim_head=64, dropout=0.0):
    return {
        "image_size": image_size, "patch_size": patch_size, "dim": dim, "depth": depth,
        "num_head": num_head, "mlp_dim": mlp_dim, channel_key: 3, "dim_head": dim_head,
        "dropout": dropout,
    }


def _vqgan(image_size, patch_size, dim, depth, num_head, mlp_dim, n_embed=8192, embed_dim=32):
    return {
        "n_embed": n_embed, "embed_dim": embed_dim, "beta": 0.25,
        "enc": _tower(image_size, patch_size, dim, depth, num_head, mlp_dim, "in_channels"),
        "dec": _tower(image_size, patch_size, dim, depth, num_head, mlp_dim, "out_channels"),
    }


# reference config.py:40-68
vit_s_vqgan_config = _vqgan(256, 8, 512, 8, 8, 2048)

# reference config.py:70-77
pipeline_v1_config = {
    "stage1": "vit-s-vqgan", "t5": "t5-l", "dim": 1024, "dim_head": 64, "mlp_dim": 4096,
    "num_head": 16, "depth": 12, "dropout": 0.1,
}

ver2cfg = {
    "vit-s-vqgan": vit_s_vqgan_config,
    "paintmindv1": pipeline_v1_config,
}

# ---- synthetic configurations named by BASELINE.json (not present in the reference) -------------
# ``context_dim`` / ``text_model`` are extension keys: when present they override the reference's
# t5 lookup table (generate.py:52-53) so that a pipeline can be built without a text tower.
ver2cfg["vit-b-vqgan-512"] = _vqgan(512, 16, 768, 12, 12, 3072)            # BASELINE cfg 5 (assumed)
ver2cfg["bench-uncond-12L-d512"] = {                                        # BASELINE cfg 3
    "stage1": "vit-s-vqgan", "t5": "t5-l", "text_model": "none", "context_dim": 512,
    "dim": 512, "dim_head": 64, "mlp_dim": 2048, "num_head": 8, "depth": 12, "dropout": 0.1,
}
ver2cfg["bench-text-24L-d768"] = {                                          # BASELINE cfg 4
    "stage1": "vit-s-vqgan", "t5": "t5-l", "text_model": "none", "context_dim": 768,
    "dim": 768, "dim_head": 64, "mlp_dim": 3072, "num_head": 12, "depth": 24, "dropout": 0.1,
}
ver2cfg["bench-text-24L-d1024-512px"] = {                                   # BASELINE cfg 5
    "stage1": "vit-b-vqgan-512", "t5": "t5-l", "text_model": "none", "context_dim": 768,
    "dim": 1024, "dim_head": 64, "mlp_dim": 4096, "num_head": 16, "depth": 24, "dropout": 0.1,
}
# tiny configurations used by the golden fixtures (tests/golden/make_goldens.py)
ver2cfg["tiny-vqgan"] = _vqgan(32, 8, 64, 2, 2, 128, n_embed=64, embed_dim=32)
ver2cfg["tiny-pipeline"] = {
    "stage1": "tiny-vqgan", "t5": "t5-l", "text_model": "none", "context_dim": 96,
    "dim": 128, "dim_head": 64, "mlp_dim": 256, "num_head": 2, "depth": 2, "dropout": 0.1,
}
