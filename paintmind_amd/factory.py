"""Model factory with the reference's call signature and error behaviour (reference paintmind/factory.py:6-26).

    create_model(arch, version, pretrained, checkpoint_path) -> VQModel | Pipeline
    create_pipeline_for_train(version, stage1_pretrained, stage1_checkpoint_path) -> Pipeline
"""
from .config import Config, ver2cfg

_HUB_OWNER = "RootYuan"          # the reference publishes its checkpoints as <owner>/<version>/<version>.pt


def _build_vqgan(config):
    from .stage1 import VQModel
    return VQModel(config)


def _build_pipeline(config):
    from .generate import Pipeline
    return Pipeline(config, stage1_pretrained=False)


_BUILDERS = {"vqgan": _build_vqgan, "pipeline": _build_pipeline}


def _default_checkpoint(version):
    """fetch <version>.pt from the hub repo named after the version (needs network access)"""
    from huggingface_hub import hf_hub_download
    return hf_hub_download(f"{_HUB_OWNER}/{version}", f"{version}.pt")


def create_model(arch='pipeline', version='paintmindv1', pretrained=True, checkpoint_path=None):
    config = Config(ver2cfg[version])           # an unknown version is a KeyError, as in the reference
    builder = _BUILDERS.get(arch)
    if builder is None:
        raise ValueError(f"failed to load arch named {arch}")
    model = builder(config)
    if pretrained:
        model.from_pretrained(checkpoint_path if checkpoint_path is not None else _default_checkpoint(version))
    return model


def create_pipeline_for_train(version='paintmindv1', stage1_pretrained=True, stage1_checkpoint_path=None):
    from .generate import Pipeline
    return Pipeline(Config(ver2cfg[version]), stage1_pretrained=stage1_pretrained,
                    stage1_checkpoint_path=stage1_checkpoint_path)
