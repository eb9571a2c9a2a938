"""Drop-in for the reference's ``src/processer/transforms.py`` without torchvision (absent here).

``get_transform(type, keep_ratio, image_size)`` (reference :4-47) returns a callable
PIL.Image -> float tensor [3, S, S]: resize (bilinear for 'clip'/'clipa' — torchvision's default —
bicubic for 'sd'), optional centre crop, scale to [0,1], normalise.

Two modes, same numbers (tests/test_preprocess.py):
* as constructed — host PIL + torch, fp32 CPU tensor out, exactly what the reference's Compose returns;
* after ``transform.to(device, dtype)`` — the raw uint8 pixels are uploaded and resize / crop / normalise / cast run as
  HIP kernels (seedstory.preprocess.DevicePreprocessor; Pillow's integer resampler restated bit-exactly), the result is
  already on the device in the model dtype so the driver's ``.to(device, dtype)`` (gen_george.py:166) is a no-op.
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
        self._dev = None

    def to(self, device, dtype=torch.bfloat16):
        from seedstory.preprocess import DevicePreprocessor
        mean, std = _NORMS[self.kind]
        self._dev = DevicePreprocessor(mean, std, self.size, keep_ratio=self.keep_ratio,
                                       filt="bicubic" if self.kind == "sd" else "bilinear", device=device, dtype=dtype)
        return self

    def __call__(self, img):
        if self._dev is not None:
            return self._dev(img)
        img = img.convert("RGB")
        S = self.size
        if self.keep_ratio:
            from seedstory.preprocess import torchvision_resize_geometry
            w, h = img.size
            nw, nh, left, top = torchvision_resize_geometry(w, h, S, True)
            img = img.resize((nw, nh), self.resample)
            img = img.crop((left, top, left + S, top + S))
        else:
            img = img.resize((S, S), self.resample)
        x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255.0
        return (x - self.mean) / self.std


def get_transform(type='clip', keep_ratio=True, image_size=224):
    return _Transform(type, keep_ratio, image_size)
