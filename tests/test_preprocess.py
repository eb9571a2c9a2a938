"""SURVEY.md §8f row 3 — device image pre-processing vs the real third-party code the reference calls: Pillow's resampler
(torchvision ``Resize`` on a PIL image is ``Image.resize``) + torch's ToTensor / Normalize arithmetic.

CPU: the tap tables of ``ss_resample_coeffs`` (host C) drive a numpy restatement of the two integer passes; the result
must equal ``PIL.Image.resize`` BYTE FOR BYTE for bilinear and bicubic, up- and down-scaling, ragged sizes.
GPU: ``ss_image_preprocess`` output uint8 == PIL exactly, and the normalised tensor == the host transform bit for bit
in fp32 (same fp32 op order), == its RNE cast in bf16."""
import numpy as np
import pytest
import torch
from PIL import Image

from seedstory import preprocess as P

DEV = "cuda:0"
PIL_FILTER = {"bilinear": Image.BILINEAR, "bicubic": Image.BICUBIC}


def _img(seed, h, w):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    smooth = (128 + 100 * np.sin(xx / 17.0)[..., None] * np.cos(yy / 11.0)[..., None] * np.ones(3)).astype(np.uint8)
    return np.where(rng.rand(h, w, 1) < 0.5, base, smooth).astype(np.uint8)


def _resample_axis(a, coef, bounds):
    """One pass along axis 1 of a [R, N, 3] uint8 array: (1 << 21) + sum(pixel * tap) >> 22, clipped."""
    out = np.zeros((a.shape[0], coef.shape[0], 3), dtype=np.uint8)
    a64 = a.astype(np.int64)
    for o in range(coef.shape[0]):
        x0, n = bounds[o]
        acc = (1 << 21) + (a64[:, x0:x0 + n, :] * coef[o, :n].astype(np.int64)[None, :, None]).sum(axis=1)
        out[:, o, :] = np.clip(acc >> 22, 0, 255)
    return out


def numpy_resize(src, ow, oh, filt):
    ch, bh = P.resample_coeffs(src.shape[1], ow, filt)
    cv, bv = P.resample_coeffs(src.shape[0], oh, filt)
    tmp = _resample_axis(src, ch, bh)
    return _resample_axis(tmp.transpose(1, 0, 2), cv, bv).transpose(1, 0, 2)


@pytest.mark.parametrize("filt", ["bilinear", "bicubic"])
@pytest.mark.parametrize("h,w,oh,ow", [(300, 200, 448, 448), (1024, 1024, 448, 448), (517, 733, 448, 448), (448, 448, 448, 448),
                                       (97, 1500, 224, 224), (64, 64, 100, 37)])
def test_tap_tables_reproduce_pillow_exactly(filt, h, w, oh, ow):
    src = _img(h * 7 + w, h, w)
    ref = np.asarray(Image.fromarray(src).resize((ow, oh), PIL_FILTER[filt]))
    got = numpy_resize(src, ow, oh, filt)
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_geometry_matches_torchvision_rules():
    assert P.torchvision_resize_geometry(300, 200, 448, False) == (448, 448, 0, 0)
    assert P.torchvision_resize_geometry(300, 200, 224, True) == (336, 224, 56, 0)
    assert P.torchvision_resize_geometry(333, 500, 224, True) == (224, 336, 0, 56)       # int(224*500/333) = 336
    assert P.torchvision_resize_geometry(1000, 333, 224, True) == (672, 224, 224, 0)     # int(672.67) truncates
    assert P.torchvision_resize_geometry(227, 224, 224, True)[2] == 2                    # round(1.5) -> 2 (half-even)
    assert P.torchvision_resize_geometry(229, 224, 224, True)[2] == 2                    # round(2.5) -> 2


def test_host_transform_is_pil_plus_torch_ops():
    from src.processer.transforms import get_transform
    src = _img(5, 200, 300)
    t = get_transform("clip", keep_ratio=False, image_size=448)
    x = t(Image.fromarray(src))
    ref = torch.from_numpy(np.asarray(Image.fromarray(src).resize((448, 448), Image.BILINEAR)).copy()).permute(2, 0, 1).float().div(255)
    ref = (ref - torch.tensor((0.48145466, 0.4578275, 0.40821073)).view(3, 1, 1)) / torch.tensor(
        (0.26862954, 0.26130258, 0.27577711)).view(3, 1, 1)
    assert torch.equal(x, ref)


@pytest.mark.gpu
@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs the MI355X (the rest of this file runs on CPU)")
@pytest.mark.parametrize("kind,keep_ratio,size,h,w", [("clip", False, 448, 300, 200), ("clip", False, 448, 1024, 1024),
                                                      ("clip", True, 224, 517, 733), ("sd", True, 256, 333, 500),
                                                      ("clipa", False, 448, 448, 448), ("sd", False, 128, 97, 1500)])
def test_device_preprocess_bit_exact(kind, keep_ratio, size, h, w):
    from src.processer.transforms import get_transform
    src = _img(h + 3 * w, h, w)
    img = Image.fromarray(src)
    host = get_transform(kind, keep_ratio=keep_ratio, image_size=size)
    ref = host(img)                                                    # PIL + torch fp32 on the CPU
    for dtype in (torch.float32, torch.bfloat16, torch.float16):
        dev = get_transform(kind, keep_ratio=keep_ratio, image_size=size).to(DEV, dtype)
        got = dev(img)
        assert got.is_cuda and got.dtype == dtype and got.shape == (3, size, size)
        assert torch.equal(got.cpu(), ref.to(dtype)), (dtype, float((got.cpu().float() - ref).abs().max()))
    y, u8 = dev._dev(img, return_u8=True)
    nw, nh, left, top = P.torchvision_resize_geometry(w, h, size, keep_ratio)
    pil = np.asarray(img.resize((nw, nh), host.resample).crop((left, top, left + size, top + size)))
    assert np.array_equal(u8.cpu().numpy(), pil)
    # the driver's `.unsqueeze(0).to(device, dtype)` on the device result is a no-op
    assert y.unsqueeze(0).to(DEV, dtype=torch.float16).data_ptr() == y.data_ptr()
