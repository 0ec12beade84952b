"""Side-by-side "origin | reconstruct" figure for a VQGAN checkpoint (what reference paintmind/reconstruct.py:23-52
produces), built on this package's transform and model."""
import io

import numpy as np
import torch
from PIL import Image, ImageDraw, ImageFont

from .factory import create_model
from .utils.transform import stage1_transform

_PANEL = 256


def restore(x):
    """CHW tensor in [-1, 1] -> 8-bit RGB PIL image (truncating, like the reference's astype(uint8))"""
    arr = ((x.detach().float().cpu() + 1) * 0.5).permute(1, 2, 0).numpy()
    return Image.fromarray((arr * 255).astype(np.uint8))


def _open(source):
    if source.startswith("http"):
        import requests
        reply = requests.get(source)
        reply.raise_for_status()
        return Image.open(io.BytesIO(reply.content))
    return Image.open(source).convert("RGB")


def _label(figure, titles):
    try:
        font = ImageFont.truetype("arialbi.ttf", 16)
    except Exception:
        font = None
    draw = ImageDraw.Draw(figure)
    for column, title in enumerate(titles):
        draw.text((column * _PANEL, 0), str(title), (255, 255, 255), font=font)


def reconstruction(img_path=None, model_name='vit-s-vqgan', titles=['origin', 'reconstruct'], checkpoint_path=None, scale=0.8,
                   device='cuda', pretrained=True):
    """encode -> decode one image and return the two panels as a single PIL image"""
    pixels = stage1_transform(is_train=False, scale=scale)(_open(img_path)).to(device)
    model = create_model(arch='vqgan', version=model_name, pretrained=pretrained, checkpoint_path=checkpoint_path).to(device).eval()
    with torch.no_grad():
        latent = model.encode(pixels[None])[0]
        rebuilt = model.decode(latent)[0]
    figure = Image.new("RGB", (2 * _PANEL, _PANEL))
    for column, panel in enumerate((pixels, rebuilt)):
        figure.paste(restore(panel), (column * _PANEL, 0))
    _label(figure, titles)
    return figure
