"""Side-by-side reconstruction figure (reference paintmind/reconstruct.py:11-52)."""
import io

import numpy as np
import torch
from PIL import Image, ImageDraw, ImageFont

from .factory import create_model
from .utils.transform import stage1_transform


def restore(x):
    """[-1,1] CHW tensor -> PIL image (reconstruct.py:11-16)"""
    x = (x + 1) * 0.5
    x = x.permute(1, 2, 0).detach().cpu().numpy()
    return Image.fromarray((255 * x).astype(np.uint8))


def download_image(url):
    import requests
    resp = requests.get(url)
    resp.raise_for_status()
    return Image.open(io.BytesIO(resp.content))


def reconstruction(img_path=None, model_name='vit-s-vqgan', titles=['origin', 'reconstruct'], checkpoint_path=None, scale=0.8,
                   device='cuda', pretrained=True):
    w, h = 256, 256
    img = download_image(img_path) if img_path.startswith('http') else Image.open(img_path).convert('RGB')
    img = stage1_transform(is_train=False, scale=scale)(img).to(device)
    model = create_model(arch='vqgan', version=model_name, pretrained=pretrained, checkpoint_path=checkpoint_path).to(device)
    model.eval()
    with torch.no_grad():
        z, _, _ = model.encode(img.unsqueeze(0))
        rec = model.decode(z).squeeze(0)
    fig = Image.new("RGB", (2 * w, h))
    fig.paste(restore(img), (0, 0))
    fig.paste(restore(rec), (w, 0))
    try:
        font = ImageFont.truetype('arialbi.ttf', 16)
    except Exception:
        font = None
    for i, title in enumerate(titles):
        ImageDraw.Draw(fig).text((i * w, 0), f'{title}', (255, 255, 255), font=font)
    return fig
