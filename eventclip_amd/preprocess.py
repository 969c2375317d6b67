"""CLIP's image ``preprocess`` on the MI355X.

The reference stores the callable returned by ``clip.load`` in
``params.data_transforms`` (test.py:29) and applies it to every frame
(datasets/event2img.py:119-122): torchvision ``Resize(n_px, BICUBIC)`` ->
``CenterCrop(n_px)`` -> RGB -> ``ToTensor`` -> ``Normalize``.  ``Preprocess`` is the
drop-in (PIL image / uint8 HWC array in, FloatTensor [3, n_px, n_px] out);
``preprocess_frames`` is the batched device entry that can also emit the 16-bit
im2col rows the patch-embedding GEMM consumes, skipping the fp32 tensor.
Integer-exact with Pillow; runs in libeventclip_hip.so, no CPU fallback.
"""
import ctypes

import numpy as np
import torch

from . import _lib

_plans = {}


def _plan(in_h, in_w, n_px, dev):
    key = (in_h, in_w, n_px, str(dev))
    if key not in _plans:
        lib = _lib.lib()
        nbytes = lib.ec_preprocess_plan_bytes(in_h, in_w, n_px)
        if nbytes == 0:
            raise ValueError(f'bad preprocess geometry {(in_h, in_w, n_px)}')
        host = np.zeros(nbytes // 4, dtype=np.int32)
        _lib.check(lib.ec_preprocess_plan(in_h, in_w, n_px, host.ctypes.data, nbytes),
                   'ec_preprocess_plan')
        _plans[key] = (host, torch.from_numpy(host).to(dev))
    return _plans[key]


def plan_geometry(in_h, in_w, n_px):
    """(new_h, new_w, top, left) of Resize + CenterCrop for this input size."""
    host, _ = _plan(in_h, in_w, n_px, _lib.require_gpu())
    return int(host[4]), int(host[5]), int(host[6]), int(host[7])


def preprocess_frames(frames, n_px=224, mode='chw', patch=None, kpad=None, dtype=torch.float16,
                      out=None):
    """frames: uint8 CUDA tensor [F, H, W, 3].

    mode 'chw'     -> float32 [F, 3, n_px, n_px]   (the reference's tensor)
    mode 'patches' -> 16-bit  [F, G, kpad]         (input of CLIP.encode_patches)
    mode 'u8'      -> uint8   [F, n_px, n_px, 3]   (resized + cropped only)
    """
    dev = _lib.require_gpu()
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 \
        and frames.shape[3] == 3 and frames.is_contiguous()
    F, H, W, _ = frames.shape
    host, plan = _plan(H, W, n_px, dev)
    code = _lib.EC_F16 if dtype == torch.float16 else _lib.EC_BF16
    if mode == 'chw':
        m, shape, odt = _lib.EC_PRE_CHW_F32, (F, 3, n_px, n_px), torch.float32
        patch, kpad = 1, 0
    elif mode == 'u8':
        m, shape, odt = _lib.EC_PRE_HWC_U8, (F, n_px, n_px, 3), torch.uint8
        patch, kpad = 1, 0
    elif mode == 'patches':
        assert patch and kpad
        g = n_px // patch
        m, shape, odt = _lib.EC_PRE_PATCHES16, (F, g * g, kpad), dtype
    else:
        raise ValueError(mode)
    if out is None:
        from . import torch_ops  # noqa: F401  (registers eventclip_hip::preprocess)
        return torch.ops.eventclip_hip.preprocess(frames, int(n_px), m, int(patch), int(kpad), code)
    assert out.dtype == odt and out.is_contiguous() and out.numel() >= int(np.prod(shape))
    rc = _lib.lib().ec_preprocess(_lib.ptr(frames), F, host.ctypes.data, _lib.ptr(plan),
                                  _lib.ptr(out), m, patch, kpad, code, _lib.stream_ptr())
    _lib.check(rc, 'ec_preprocess')
    return out


class Preprocess:
    """Drop-in for the ``preprocess`` callable of ``clip.load`` (one image at a time)."""

    def __init__(self, n_px):
        self.n_px = int(n_px)

    def __call__(self, img):
        dev = _lib.require_gpu()
        if isinstance(img, torch.Tensor):
            arr = img
        else:
            if hasattr(img, 'convert'):          # PIL image
                img = img.convert('RGB')
            arr = torch.from_numpy(np.ascontiguousarray(np.asarray(img)))
        if arr.dtype != torch.uint8 or arr.dim() != 3 or arr.shape[2] != 3:
            raise TypeError('preprocess expects an RGB uint8 image [H, W, 3]')
        out = preprocess_frames(arr.to(dev).contiguous()[None], self.n_px, mode='chw')
        return out[0]

    def __repr__(self):
        return (f'Preprocess(Resize({self.n_px}, bicubic) -> CenterCrop({self.n_px}) -> '
                'ToTensor -> Normalize(CLIP mean/std)) [HIP]')
