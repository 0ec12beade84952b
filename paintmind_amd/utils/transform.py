"""Image preprocessing for the tokenizer (reference paintmind/utils/transform.py:7-34), without torchvision.

The reference composes torchvision transforms on PIL images: bicubic Resize to int(img_size/scale) square,
RandomCrop(+RandomHorizontalFlip for stage 1) in training or CenterCrop in evaluation, ToTensor, Normalize(0.5, 0.5)
-> a float tensor in [-1, 1].  On PIL inputs torchvision's Resize is PIL's own resize, so the same sequence is
restated here with PIL + torch only (random draws use the torch generator in torchvision's order: crop row,
crop column, then the flip coin)."""
import numpy as np
import torch
from PIL import Image


def pair(t):
    return t if isinstance(t, tuple) else (t, t)


class _Pipeline:
    def __init__(self, img_size, is_train, scale, flip):
        self.img_size, self.is_train, self.flip = img_size, is_train, flip
        self.resize = pair(int(img_size / scale))

    def __call__(self, img):
        if not isinstance(img, Image.Image):
            raise TypeError("expected a PIL image")
        h, w = self.resize
        img = img.resize((w, h), Image.BICUBIC)
        th = tw = self.img_size
        if self.is_train:
            top = int(torch.randint(0, h - th + 1, size=(1,)).item())
            left = int(torch.randint(0, w - tw + 1, size=(1,)).item())
        else:
            top, left = int(round((h - th) / 2.0)), int(round((w - tw) / 2.0))
        img = img.crop((left, top, left + tw, top + th))
        if self.is_train and self.flip and torch.rand(1) < 0.5:
            img = img.transpose(Image.FLIP_LEFT_RIGHT)
        arr = np.asarray(img, dtype=np.uint8)
        if arr.ndim == 2:
            arr = arr[:, :, None]
        x = torch.from_numpy(arr.copy()).permute(2, 0, 1).float().div(255)
        return (x - 0.5) / 0.5


def stage1_transform(img_size=256, is_train=True, scale=0.8):
    return _Pipeline(img_size, is_train, scale, flip=True)


def stage2_transform(img_size=256, is_train=True, scale=0.8):
    return _Pipeline(img_size, is_train, scale, flip=False)
