"""Stage 2, the MaskGIT / Muse-style bidirectional transformer over the 32x32 token grid (reference paintmind/stage2)."""
from .transformer import CondTransformer, Layer

__all__ = ["CondTransformer", "Layer"]
