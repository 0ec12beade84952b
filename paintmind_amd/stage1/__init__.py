"""Stage 1, the ViT-VQGAN tokenizer (reference paintmind/stage1): encoder / decoder towers, the L2-normalised
codebook and the VQModel facade.  Only VQModel is re-exported by the reference; the building blocks are exported
here as well because they are the operator plug-points of INTEGRATION.md."""
from .layers import Decoder, Encoder, Layer, Transformer
from .quantize import VectorQuantizer
from .vqmodel import VQModel

__all__ = ["VQModel", "Encoder", "Decoder", "Transformer", "Layer", "VectorQuantizer"]
