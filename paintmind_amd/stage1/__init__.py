from .vqmodel import VQModel
