"""Drop-in for the reference's ``src/processer/transforms.py`` without torchvision (absent here).

``get_transform(type, keep_ratio, image_size)`` (reference :4-47) returns a callable
PIL.Image -> float tensor [3, S, S]: resize (bilinear for 'clip'/'clipa' — torchvision's default —
bicubic for 'sd'), optional centre crop, scale to [0,1], normalise.  Host-side pre-processing
(one 448x448 image per story); not a GPU hot spot.
"""
import numpy as np
import torch
from PIL import Image

_NORMS = {
    "clip": ((0.48145466, 0.4578275, 0.40821073), (0.26862954, 0.26130258, 0.27577711)),
    "clipa": ((0.485, 0.456, 0.406), (0.229, 0.224, 0.225)),
    "sd": ((0.5, 0.5, 0.5), (0.5, 0.5, 0.5)),
}


class _Transform:
    def __init__(self, kind, keep_ratio, size):
        if kind not in _NORMS:
            raise NotImplementedError(kind)
        self.kind, self.keep_ratio, self.size = kind, keep_ratio, size
        self.resample = Image.BICUBIC if kind == "sd" else Image.BILINEAR
        mean, std = _NORMS[kind]
        self.mean = torch.tensor(mean).view(3, 1, 1)
        self.std = torch.tensor(std).view(3, 1, 1)

    def __call__(self, img):
        img = img.convert("RGB")
        S = self.size
        if self.keep_ratio:
            w, h = img.size
            if w <= h:
                nw, nh = S, max(S, int(round(h * S / w)))
            else:
                nw, nh = max(S, int(round(w * S / h))), S
            img = img.resize((nw, nh), self.resample)
            left, top = (nw - S) // 2, (nh - S) // 2
            img = img.crop((left, top, left + S, top + S))
        else:
            img = img.resize((S, S), self.resample)
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        return (x - self.mean) / self.std


def get_transform(type='clip', keep_ratio=True, image_size=224):
    return _Transform(type, keep_ratio, image_size)
