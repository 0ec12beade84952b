"""paintmind_amd -- MI355X-native generation path of PaintMind behind the reference's Python API.

    import paintmind_amd as pm
    model = pm.create_model(arch='vqgan', version='vit-s-vqgan', pretrained=False).to('cuda')
    z, loss, idx = model.encode(x); rec = model.decode(z)

Mirrors reference paintmind/__init__.py:1-7 for the inference surface (Config, create_model,
create_pipeline_for_train) plus the image I/O either side of the path (stage1/2_transform, reconstruction:
SURVEY.md section 8(f) row 3).  The trainers of the reference are out of scope.
"""
from .version import __version__
from .config import Config, ver2cfg
from .factory import create_model, create_pipeline_for_train
from .utils.transform import stage1_transform, stage2_transform
from .reconstruct import reconstruction

__all__ = ["__version__", "Config", "ver2cfg", "create_model", "create_pipeline_for_train", "stage1_transform",
           "stage2_transform", "reconstruction"]
