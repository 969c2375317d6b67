"""Oracle for CLIP's image preprocess (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference applies ``params.data_transforms`` = the ``preprocess`` returned by
``clip.load`` to every frame (/root/reference/datasets/event2img.py:119-122,
test.py:26-29).  That callable lives in un-vendored openai/CLIP
(clip/clip.py ``_transform``): torchvision ``Resize(n_px, BICUBIC)`` ->
``CenterCrop(n_px)`` -> RGB -> ``ToTensor`` -> ``Normalize(mean, std)`` (the
constants also appear at /root/reference/method.py:17-18).  torchvision hands a
PIL image to ``Image.resize``, so the arithmetic that matters is Pillow's
``src/libImaging/Resample.c`` 8-bit path, restated here in numpy:

* separable two-pass resample, horizontal first, uint8 rounding between passes;
* coefficients in double (bicubic a = -0.5, support scaled by the downscale
  factor), each window normalised to sum 1, then fixed point with
  PRECISION_BITS = 22 and round-half-away-from-zero;
* accumulate from 1 << 21, arithmetic shift by 22, clamp to [0, 255].

Pinned: bit-exact against ``PIL.Image.resize(..., BICUBIC)`` (tests run that
comparison live, Pillow is installed wherever the tests run).
"""
import math

import numpy as np

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
PRECISION_BITS = 32 - 8 - 2


def _bicubic(x):
    a = -0.5
    x = abs(x)
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size, out_size):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the full box [0, in_size).
    Returns bounds int32 [out, 2] (xmin, count) and kk int32 [out, ksize]."""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            if v < 0:
                kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS))
            else:
                kk[xx, x] = int(0.5 + v * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img, bounds, kk, axis):
    """One 8-bit pass along ``axis`` of an [H, W, C] uint8 image."""
    src = np.moveaxis(img, axis, 0).astype(np.int64)          # [in, other, C]
    out_size, ksize = kk.shape
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(cnt):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resized_size(h, w, n_px):
    """torchvision Resize(int): short side -> n_px, long side = int(n_px * long / short)."""
    if w <= h:
        return int(n_px * h / w), n_px          # (new_h, new_w)
    return n_px, int(n_px * w / h)


def resize_bicubic(img, new_h, new_w):
    """PIL Image.resize((new_w, new_h), BICUBIC) on an [H, W, 3] uint8 array."""
    h, w, _ = img.shape
    out = img
    if new_w != w:                                # horizontal pass first (ImagingResample)
        b, k = precompute_coeffs(w, new_w)
        out = _resample_axis(out, b, k, axis=1)
    if new_h != h:
        b, k = precompute_coeffs(h, new_h)
        out = _resample_axis(out, b, k, axis=0)
    return out


def center_crop_offsets(h, w, n_px):
    """torchvision center_crop: int(round((size - crop) / 2.0)), Python round-half-even."""
    return int(round((h - n_px) / 2.0)), int(round((w - n_px) / 2.0))


def normalise_lut():
    """float32 table [3, 256]: ToTensor (/255) then Normalize, in torch's operation order."""
    v = np.arange(256, dtype=np.float32) / np.float32(255)
    mean = np.array(CLIP_MEAN, dtype=np.float32)[:, None]
    std = np.array(CLIP_STD, dtype=np.float32)[:, None]
    return ((v[None, :] - mean) / std).astype(np.float32)


def resize_crop_u8(frames, n_px=224):
    """uint8 [F, H, W, 3] -> uint8 [F, n_px, n_px, 3] (Resize + CenterCrop)."""
    frames = np.asarray(frames)
    F, H, W, _ = frames.shape
    nh, nw = resized_size(H, W, n_px)
    top, left = center_crop_offsets(nh, nw, n_px)
    if nh < n_px or nw < n_px:
        raise ValueError('image smaller than the crop')
    out = np.empty((F, n_px, n_px, 3), dtype=np.uint8)
    for f in range(F):
        r = resize_bicubic(frames[f], nh, nw)
        out[f] = r[top:top + n_px, left:left + n_px]
    return out


def preprocess(frames, n_px=224):
    """CLIP ``preprocess`` over a stack of frames: uint8 [F, H, W, 3] -> float32 [F, 3, n_px, n_px]."""
    u8 = resize_crop_u8(frames, n_px)
    lut = normalise_lut()
    out = np.empty((u8.shape[0], 3, n_px, n_px), dtype=np.float32)
    for c in range(3):
        out[:, c] = lut[c][u8[..., c]]
    return out


def patchify(img, patch, kpad=None):
    """float [N, 3, R, R] -> [N, G, kpad] im2col rows in conv-weight order (c, i, j), zero padded."""
    img = np.asarray(img)
    N, C, R, _ = img.shape
    g = R // patch
    k = C * patch * patch
    kpad = kpad or ((k + 63) // 64) * 64
    x = img.reshape(N, C, g, patch, g, patch).transpose(0, 2, 4, 1, 3, 5).reshape(N, g * g, k)
    out = np.zeros((N, g * g, kpad), dtype=img.dtype)
    out[:, :, :k] = x
    return out


def patchify_split(img, patch, kpad, dtype):
    """The EC_PRE_PATCHES16 layout of include/eventclip_hip.h as a torch tensor of `dtype`
    (torch.float16 / torch.bfloat16): row = [hi | lo | 0], hi = round16(v), lo = round16(v - hi)."""
    import torch
    k = 3 * patch * patch
    x = torch.from_numpy(patchify(np.asarray(img, dtype=np.float32), patch, k))
    hi = x.to(dtype)
    lo = (x - hi.float()).to(dtype)
    out = torch.zeros(x.shape[0], x.shape[1], kpad, dtype=dtype)
    out[:, :, :k] = hi
    out[:, :, k:2 * k] = lo
    return out
