from .transformer import CondTransformer
