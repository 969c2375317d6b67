"""Frame-space RandAugment with the reference's interface (datasets/augment.py), applied on the GPU.

``RandAugment(num_ops, interpolation, fill)`` mirrors the reference's class: ``randomize_ops``
makes the same ``torch.randint`` draws in the same order (:142-157), ``forward`` applies the
same sampled ops to every view of a sample (:159-178).  The operators run in
``ec_randaugment`` (csrc/randaugment.hip), bit-exact with the Pillow calls torchvision's PIL
branch makes for the reference; only BICUBIC interpolation is built (what the reference
configures, datasets/event2img.py:36-42).  No CPU fallback.
"""
import ctypes
import math

import numpy as np
import torch

from . import _lib

# order of _augmentation_space (augment.py:123-140): the op index is a torch.randint over it
OP_NAMES = ('Identity', 'ShearX', 'ShearY', 'TranslateX', 'TranslateY', 'Rotate', 'Brightness', 'Color',
            'Contrast', 'Sharpness', 'Posterize', 'Solarize', 'AutoContrast', 'Equalize')
_SIGNED = frozenset(OP_NAMES[1:10])
_NUM_BINS = 30                                                            # augment.py:146


def _magnitudes(op, resolution):
    H, W = resolution
    n = _NUM_BINS
    if op in ('ShearX', 'ShearY'):
        return torch.linspace(0.0, 0.3, n)
    if op == 'TranslateX':
        return torch.linspace(0.0, 150.0 / 331.0 * W, n)
    if op == 'TranslateY':
        return torch.linspace(0.0, 150.0 / 331.0 * H, n)
    if op == 'Rotate':
        return torch.linspace(0.0, 30.0, n)
    if op in ('Brightness', 'Color', 'Contrast', 'Sharpness'):
        return torch.linspace(0.0, 0.9, n)
    if op == 'Posterize':
        return 8 - (torch.arange(n) / ((n - 1) / 4)).round().int()
    if op == 'Solarize':
        return torch.linspace(255.0, 0.0, n)
    return torch.tensor(0.0)


def _inverse_affine(center, translate, shear_deg):
    """torchvision F.affine's output->input matrix for angle 0, scale 1 (the shear / translate ops)."""
    sx, sy = math.radians(shear_deg[0]), math.radians(shear_deg[1])
    cx, cy = center
    tx, ty = translate
    a = math.cos(-sy) / math.cos(sy)
    b = -math.cos(-sy) * math.tan(sx) / math.cos(sy) - math.sin(0.0)
    c = math.sin(-sy) / math.cos(sy)
    d = -math.sin(-sy) * math.tan(sx) / math.cos(sy) + math.cos(0.0)
    m = [d, -b, 0.0, -c, a, 0.0]
    m[2] += m[0] * (-cx - tx) + m[1] * (-cy - ty)
    m[5] += m[3] * (-cx - tx) + m[4] * (-cy - ty)
    m[2] += cx
    m[5] += cy
    return m


def op_descriptor(name, magnitude, resolution):
    """(op name, magnitude) as augment.py:10-87 interprets it -> EcAugOp for frames of `resolution`."""
    H, W = resolution
    d = _lib.EcAugOp()
    d.kind, d.alpha, d.param = _lib.EC_AUG_IDENTITY, 1.0, 0.0
    m = None
    if name == 'Identity':
        pass
    elif name == 'ShearX':
        m = _inverse_affine([0, 0], [0, 0], [math.degrees(math.atan(magnitude)), 0.0])
    elif name == 'ShearY':
        m = _inverse_affine([0, 0], [0, 0], [0.0, math.degrees(math.atan(magnitude))])
    elif name == 'TranslateX':
        m = _inverse_affine([W * 0.5, H * 0.5], [int(magnitude), 0], [0.0, 0.0])
    elif name == 'TranslateY':
        m = _inverse_affine([W * 0.5, H * 0.5], [0, int(magnitude)], [0.0, 0.0])
    elif name == 'Rotate':                        # PIL.Image.rotate(angle, BICUBIC, expand=False)
        angle = magnitude % 360.0
        if angle == 0:
            pass
        elif angle == 180:
            d.kind = _lib.EC_AUG_ROT180
        elif angle in (90, 270) and H == W:
            d.kind = _lib.EC_AUG_ROT90 if angle == 90 else _lib.EC_AUG_ROT270
        else:
            cx, cy = W / 2.0, H / 2.0
            ang = -math.radians(angle)
            m = [round(math.cos(ang), 15), round(math.sin(ang), 15), 0.0,
                 round(-math.sin(ang), 15), round(math.cos(ang), 15), 0.0]
            m[2] = m[0] * (-cx) + m[1] * (-cy) + m[2]
            m[5] = m[3] * (-cx) + m[4] * (-cy) + m[5]
            m[2] += cx
            m[5] += cy
    elif name in ('Brightness', 'Color', 'Contrast', 'Sharpness'):
        d.kind = {'Brightness': _lib.EC_AUG_BRIGHTNESS, 'Color': _lib.EC_AUG_COLOR,
                  'Contrast': _lib.EC_AUG_CONTRAST, 'Sharpness': _lib.EC_AUG_SHARPNESS}[name]
        d.alpha = 1.0 + magnitude                 # stored as C float, as Image.blend receives it
    elif name == 'Posterize':
        d.kind, d.param = _lib.EC_AUG_POSTERIZE, float(int(magnitude))
    elif name == 'Solarize':
        d.kind, d.param = _lib.EC_AUG_SOLARIZE, float(magnitude)
    elif name == 'AutoContrast':
        d.kind = _lib.EC_AUG_AUTOCONTRAST
    elif name == 'Equalize':
        d.kind = _lib.EC_AUG_EQUALIZE
    else:
        raise ValueError(f'The provided operator {name} is not recognized.')
    if m is not None:
        d.kind = _lib.EC_AUG_AFFINE
        for i in range(6):
            d.m[i] = float(m[i])
    return d


def apply_ops(frames, per_frame_ops, fill):
    """frames uint8 CUDA [F, H, W, 3]; per_frame_ops: list (len F) of [(name, magnitude), ...] of equal
    length; fill: 3 ints.  Returns the augmented frames (new tensor)."""
    dev = _lib.require_gpu()
    assert frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4 and frames.shape[3] == 3
    frames = frames.contiguous()
    F, H, W, _ = frames.shape
    assert len(per_frame_ops) == F
    if F == 0:
        return frames.clone()
    num_ops = len(per_frame_ops[0])
    # one descriptor blob per distinct operator list (the views of a sample share theirs; filling a ctypes array
    # element by element cost more host time than the kernels take)
    cache, blobs = {}, {}

    def blob(ops):
        b = blobs.get(id(ops))
        if b is None:
            assert len(ops) == num_ops
            arr = (_lib.EcAugOp * num_ops)()
            for k, (name, mag) in enumerate(ops):
                key = (name, mag)
                if key not in cache:
                    cache[key] = op_descriptor(name, mag, (H, W))
                arr[k] = cache[key]
            b = blobs[id(ops)] = bytes(arr)
        return b
    host = np.frombuffer(b''.join([blob(ops) for ops in per_frame_ops]), dtype=np.uint8).copy()
    ops_d = torch.from_numpy(host).to(dev)
    out = torch.empty_like(frames)
    need = int(_lib.lib().ec_randaugment_workspace_bytes(F, H, W, num_ops))
    ws = torch.empty((need,), dtype=torch.uint8, device=dev)
    fill_c = (ctypes.c_uint8 * 3)(*[int(v) for v in fill])
    rc = _lib.lib().ec_randaugment(_lib.ptr(frames), _lib.ptr(out), F, H, W, _lib.ptr(ops_d), num_ops,
                                   ctypes.cast(fill_c, ctypes.c_void_p), _lib.ptr(ws), need,
                                   _lib.stream_ptr())
    _lib.check(rc, 'ec_randaugment')
    return out


class RandAugment(torch.nn.Module):
    """The reference's RandAugment (augment.py:90-193): ``num_ops`` operators drawn per call of
    ``forward`` and applied, the same ones, to every image of the list / every frame of the tensor."""

    def __init__(self, num_ops=2, interpolation='bicubic', fill=None):
        super().__init__()
        if str(getattr(interpolation, 'value', interpolation)).lower() != 'bicubic':
            raise NotImplementedError('only BICUBIC interpolation is built (event2img.py:36-42)')
        self.num_ops = num_ops
        self.interpolation = interpolation
        self.fill = fill
        self.cur_ops = None

    def randomize_ops(self, resolution):
        """Randomly select `self.num_ops` augmentations to apply (augment.py:142-157)."""
        assert self.cur_ops is None, 'Unused RandAugment ops'
        self.cur_ops = []
        cur_magnitude = int(torch.randint(_NUM_BINS, (1,)).item())
        for _ in range(self.num_ops):
            op_index = int(torch.randint(len(OP_NAMES), (1,)).item())
            op_name = OP_NAMES[op_index]
            magnitudes = _magnitudes(op_name, resolution)
            magnitude = float(magnitudes[cur_magnitude].item()) if magnitudes.ndim > 0 else 0.0
            if op_name in _SIGNED and torch.randint(2, (1,)):
                magnitude *= -1.0
            self.cur_ops.append((op_name, magnitude))

    def _fill(self):
        fill = self.fill
        if fill is None:
            return (0, 0, 0)                       # PIL's default outside colour
        if isinstance(fill, (int, float)):
            return (int(fill),) * 3
        return tuple(int(f) for f in fill)

    def forward(self, imgs):
        """imgs: uint8 CUDA tensor [T, H, W, 3] (the views of one sample), or a list of [H, W, 3]
        tensors / arrays / PIL images.  Returns the same container kind with the ops applied."""
        dev = _lib.require_gpu()
        as_list = not torch.is_tensor(imgs)
        if as_list:
            pil = [hasattr(i, 'convert') for i in imgs]
            frames = torch.stack([torch.from_numpy(np.ascontiguousarray(np.asarray(i))) for i in imgs]).to(dev)
        else:
            frames = imgs
        T, H, W, _ = frames.shape
        self.randomize_ops((H, W))
        out = apply_ops(frames, [self.cur_ops] * T, self._fill())
        self.cur_ops = None
        if not as_list:
            return out
        host = out.cpu().numpy()
        if all(pil):
            from PIL import Image
            return [Image.fromarray(h) for h in host]
        return [torch.from_numpy(h) for h in host]
