"""Device image pre-processing: PIL-exact resize (+ centre crop) + ToTensor + Normalize + cast as
two HIP kernels (csrc/ss_image.hip).  The host only computes the two tap tables (Pillow's coefficient pre-computation,
``ss_resample_coeffs``) once per (source size, target size) and uploads the raw uint8 pixels; the [3, S, S] ViT input is
produced in HBM in the model dtype — the reference does resize / normalise on the host in fp32 and ships 3x4 bytes per
pixel over PCIe (transforms.py:4-19, gen_george.py:166)."""
import ctypes as C

import numpy as np
import torch

from . import _lib

FILTERS = {"bilinear": _lib.SS_FILTER_BILINEAR, "bicubic": _lib.SS_FILTER_BICUBIC}
_DT = {torch.float32: _lib.SS_F32, torch.bfloat16: _lib.SS_BF16, torch.float16: _lib.SS_F16}


def resample_coeffs(in_size, out_size, filt):
    """(coef [out, ksize] int32, bounds [out, 2] int32) of one axis — host arithmetic, no GPU needed."""
    lib = _lib.lib()
    f = FILTERS[filt] if isinstance(filt, str) else filt
    ksize = lib.ss_resample_ksize(in_size, out_size, f)
    if ksize <= 0:
        raise _lib.SSError("bad resample geometry %s -> %s" % (in_size, out_size))
    coef = np.zeros((out_size, ksize), dtype=np.int32)
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    _lib.check(lib.ss_resample_coeffs(in_size, out_size, f, coef.ctypes.data_as(_lib.i32p), bounds.ctypes.data_as(_lib.i32p)),
               "ss_resample_coeffs")
    return coef, bounds


def torchvision_resize_geometry(w, h, size, keep_ratio):
    """(new_w, new_h, crop_left, crop_top) of ``Resize(size)`` + ``CenterCrop(size)`` (keep_ratio) or
    ``Resize((size, size))`` — torchvision's integer arithmetic: long side = int(size * long / short) (truncation),
    crop offset = int(round((dim - size) / 2.0)) (round-half-even)."""
    if not keep_ratio:
        return size, size, 0, 0
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
    return nw, nh, int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))


class DevicePreprocessor:
    def __init__(self, mean, std, size, keep_ratio=False, filt="bilinear", device="cuda:0", dtype=torch.bfloat16):
        self.mean = (C.c_float * 3)(*mean)
        self.std = (C.c_float * 3)(*std)
        self.size, self.keep_ratio, self.filt = size, keep_ratio, filt
        self.device, self.dtype = torch.device(device), dtype
        self._tables = {}

    def _axis(self, n_in, n_out):
        key = (n_in, n_out)
        if key not in self._tables:
            coef, bounds = resample_coeffs(n_in, n_out, self.filt)
            self._tables[key] = (torch.from_numpy(coef).to(self.device), torch.from_numpy(bounds).to(self.device), coef.shape[1],
                                 bounds)
        return self._tables[key]

    def __call__(self, img, return_u8=False):
        """img: PIL image (converted to RGB) or uint8 tensor/array [H, W, 3].  Returns [3, S, S] on the device."""
        if hasattr(img, "convert"):
            img = np.array(img.convert("RGB"), dtype=np.uint8)
        src = torch.as_tensor(img)
        if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3:
            raise _lib.SSError("expected a uint8 [H, W, 3] image")
        src = src.contiguous().to(self.device, non_blocking=True)
        H, W = int(src.shape[0]), int(src.shape[1])
        S = self.size
        OW, OH, left, top = torchvision_resize_geometry(W, H, S, self.keep_ratio)
        ch, bh, kh, _ = self._axis(W, OW)
        cv, bv, kv, bv_host = self._axis(H, OH)
        first = int(bv_host[top, 0])
        last = int(bv_host[top + S - 1, 0] + bv_host[top + S - 1, 1])
        tmp = torch.empty((last - first) * OW * 3, dtype=torch.uint8, device=self.device)
        dst = torch.empty(3, S, S, dtype=self.dtype, device=self.device)
        u8 = torch.empty(S, S, 3, dtype=torch.uint8, device=self.device) if return_u8 else None
        from . import ops
        _lib.check(_lib.lib().ss_image_preprocess(src.data_ptr(), H, W, dst.data_ptr(), u8.data_ptr() if return_u8 else None, OH, OW,
                                                  top, left, S, S, ch.data_ptr(), bh.data_ptr(), kh, cv.data_ptr(), bv.data_ptr(),
                                                  kv, first, last - first, tmp.data_ptr(), self.mean, self.std,
                                                  _DT[self.dtype], ops.stream()), "ss_image_preprocess")
        return (dst, u8) if return_u8 else dst
